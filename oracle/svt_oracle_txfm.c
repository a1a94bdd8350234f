/*
 * oracle/svt_oracle_txfm.c - TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 * CPU restatement of the residual / transform / quantisation / distortion / SATD leaf
 * kernels of the EncDec half.  Citations: /root/reference/Source/Lib/...
 *
 * The forward/inverse transforms of the reference are HM-style partial butterflies
 * (C_DEFAULT/EbTransforms_C.c:269-1600).  Restated here once, generically:
 *   forward N-point:  level-m "odd" vector o_m[j] = e_{m-1}[j] - e_{m-1}[n_m-1-j],
 *                     "even" vector e_m[j] = e_{m-1}[j] + e_{m-1}[n_m-1-j]  (e_{-1} = input);
 *                     output k = odd*2^m uses o_m with matrix row k, outputs 0 and N/2 use the
 *                     last even pair.  The low-precision "Estimate" variants keep the first
 *                     (16-point) / first two (32-point) levels in 16-bit, i.e. they WRAP
 *                     (EB_S16 even/odd/evenEven/evenOdd, :492-520, :827-840).
 *   inverse:          exact integer matrix product, (sum + offset) >> shift, clip to 16 bits.
 */
#include <stdlib.h>
#include "svt_oracle.h"

/* HEVC core transform matrix (ITU-T H.265 8.6.4.2, transMatrix); the 16/8/4-point matrices
 * are its even rows (DctCoef32x32/16x16/8x8/4x4, C_DEFAULT/EbTransforms_C.c:12-86). */
static const int8_t T32[32][32] = {
    {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64},
    {90, 90, 88, 85, 82, 78, 73, 67, 61, 54, 46, 38, 31, 22, 13, 4, -4, -13, -22, -31, -38, -46, -54, -61, -67, -73, -78, -82, -85, -88, -90, -90},
    {90, 87, 80, 70, 57, 43, 25, 9, -9, -25, -43, -57, -70, -80, -87, -90, -90, -87, -80, -70, -57, -43, -25, -9, 9, 25, 43, 57, 70, 80, 87, 90},
    {90, 82, 67, 46, 22, -4, -31, -54, -73, -85, -90, -88, -78, -61, -38, -13, 13, 38, 61, 78, 88, 90, 85, 73, 54, 31, 4, -22, -46, -67, -82, -90},
    {89, 75, 50, 18, -18, -50, -75, -89, -89, -75, -50, -18, 18, 50, 75, 89, 89, 75, 50, 18, -18, -50, -75, -89, -89, -75, -50, -18, 18, 50, 75, 89},
    {88, 67, 31, -13, -54, -82, -90, -78, -46, -4, 38, 73, 90, 85, 61, 22, -22, -61, -85, -90, -73, -38, 4, 46, 78, 90, 82, 54, 13, -31, -67, -88},
    {87, 57, 9, -43, -80, -90, -70, -25, 25, 70, 90, 80, 43, -9, -57, -87, -87, -57, -9, 43, 80, 90, 70, 25, -25, -70, -90, -80, -43, 9, 57, 87},
    {85, 46, -13, -67, -90, -73, -22, 38, 82, 88, 54, -4, -61, -90, -78, -31, 31, 78, 90, 61, 4, -54, -88, -82, -38, 22, 73, 90, 67, 13, -46, -85},
    {83, 36, -36, -83, -83, -36, 36, 83, 83, 36, -36, -83, -83, -36, 36, 83, 83, 36, -36, -83, -83, -36, 36, 83, 83, 36, -36, -83, -83, -36, 36, 83},
    {82, 22, -54, -90, -61, 13, 78, 85, 31, -46, -90, -67, 4, 73, 88, 38, -38, -88, -73, -4, 67, 90, 46, -31, -85, -78, -13, 61, 90, 54, -22, -82},
    {80, 9, -70, -87, -25, 57, 90, 43, -43, -90, -57, 25, 87, 70, -9, -80, -80, -9, 70, 87, 25, -57, -90, -43, 43, 90, 57, -25, -87, -70, 9, 80},
    {78, -4, -82, -73, 13, 85, 67, -22, -88, -61, 31, 90, 54, -38, -90, -46, 46, 90, 38, -54, -90, -31, 61, 88, 22, -67, -85, -13, 73, 82, 4, -78},
    {75, -18, -89, -50, 50, 89, 18, -75, -75, 18, 89, 50, -50, -89, -18, 75, 75, -18, -89, -50, 50, 89, 18, -75, -75, 18, 89, 50, -50, -89, -18, 75},
    {73, -31, -90, -22, 78, 67, -38, -90, -13, 82, 61, -46, -88, -4, 85, 54, -54, -85, 4, 88, 46, -61, -82, 13, 90, 38, -67, -78, 22, 90, 31, -73},
    {70, -43, -87, 9, 90, 25, -80, -57, 57, 80, -25, -90, -9, 87, 43, -70, -70, 43, 87, -9, -90, -25, 80, 57, -57, -80, 25, 90, 9, -87, -43, 70},
    {67, -54, -78, 38, 85, -22, -90, 4, 90, 13, -88, -31, 82, 46, -73, -61, 61, 73, -46, -82, 31, 88, -13, -90, -4, 90, 22, -85, -38, 78, 54, -67},
    {64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64},
    {61, -73, -46, 82, 31, -88, -13, 90, -4, -90, 22, 85, -38, -78, 54, 67, -67, -54, 78, 38, -85, -22, 90, 4, -90, 13, 88, -31, -82, 46, 73, -61},
    {57, -80, -25, 90, -9, -87, 43, 70, -70, -43, 87, 9, -90, 25, 80, -57, -57, 80, 25, -90, 9, 87, -43, -70, 70, 43, -87, -9, 90, -25, -80, 57},
    {54, -85, -4, 88, -46, -61, 82, 13, -90, 38, 67, -78, -22, 90, -31, -73, 73, 31, -90, 22, 78, -67, -38, 90, -13, -82, 61, 46, -88, 4, 85, -54},
    {50, -89, 18, 75, -75, -18, 89, -50, -50, 89, -18, -75, 75, 18, -89, 50, 50, -89, 18, 75, -75, -18, 89, -50, -50, 89, -18, -75, 75, 18, -89, 50},
    {46, -90, 38, 54, -90, 31, 61, -88, 22, 67, -85, 13, 73, -82, 4, 78, -78, -4, 82, -73, -13, 85, -67, -22, 88, -61, -31, 90, -54, -38, 90, -46},
    {43, -90, 57, 25, -87, 70, 9, -80, 80, -9, -70, 87, -25, -57, 90, -43, -43, 90, -57, -25, 87, -70, -9, 80, -80, 9, 70, -87, 25, 57, -90, 43},
    {38, -88, 73, -4, -67, 90, -46, -31, 85, -78, 13, 61, -90, 54, 22, -82, 82, -22, -54, 90, -61, -13, 78, -85, 31, 46, -90, 67, 4, -73, 88, -38},
    {36, -83, 83, -36, -36, 83, -83, 36, 36, -83, 83, -36, -36, 83, -83, 36, 36, -83, 83, -36, -36, 83, -83, 36, 36, -83, 83, -36, -36, 83, -83, 36},
    {31, -78, 90, -61, 4, 54, -88, 82, -38, -22, 73, -90, 67, -13, -46, 85, -85, 46, 13, -67, 90, -73, 22, 38, -82, 88, -54, -4, 61, -90, 78, -31},
    {25, -70, 90, -80, 43, 9, -57, 87, -87, 57, -9, -43, 80, -90, 70, -25, -25, 70, -90, 80, -43, -9, 57, -87, 87, -57, 9, 43, -80, 90, -70, 25},
    {22, -61, 85, -90, 73, -38, -4, 46, -78, 90, -82, 54, -13, -31, 67, -88, 88, -67, 31, 13, -54, 82, -90, 78, -46, 4, 38, -73, 90, -85, 61, -22},
    {18, -50, 75, -89, 89, -75, 50, -18, -18, 50, -75, 89, -89, 75, -50, 18, 18, -50, 75, -89, 89, -75, 50, -18, -18, 50, -75, 89, -89, 75, -50, 18},
    {13, -38, 61, -78, 88, -90, 85, -73, 54, -31, 4, 22, -46, 67, -82, 90, -90, 82, -67, 46, -22, -4, 31, -54, 73, -85, 90, -88, 78, -61, 38, -13},
    {9, -25, 43, -57, 70, -80, 87, -90, 90, -87, 80, -70, 57, -43, 25, -9, -9, 25, -43, 57, -70, 80, -87, 90, -90, 87, -80, 70, -57, 43, -25, 9},
    {4, -13, 22, -31, 38, -46, 54, -61, 67, -73, 78, -82, 85, -88, 90, -90, 90, -90, 88, -85, 82, -78, 73, -67, 61, -54, 46, -38, 31, -22, 13, -4}};

/* N-point matrix entry: row k, column j of the N x N DCT = T32[k * 32/N][j] (j < N) */
static inline int tcoef(int n, int k, int j) { return T32[k * (32 / n)][j]; }

/* 1-D forward partial butterfly over `n` rows; out is transposed (out[k*dstStride + row]).
 * wrap_levels: number of leading levels whose even/odd vectors are kept in int16. */
static void fwd_1d(const int16_t *in, int srcStride, int16_t *out, int dstStride, int n, int shift, int wrap_levels)
{
    const int16_t offset = (int16_t)(1 << (shift - 1));
    for (int r = 0; r < n; r++) {
        int32_t e[32], o[32];
        for (int j = 0; j < n; j++)
            e[j] = in[r * srcStride + j];
        int len = n, level = 0;
        while (len > 2) {
            const int half = len >> 1;
            int32_t ne[16];
            for (int j = 0; j < half; j++) {
                int32_t s = e[j] + e[len - 1 - j], d = e[j] - e[len - 1 - j];
                if (level < wrap_levels)
                    s = (int16_t)s, d = (int16_t)d;
                ne[j] = s, o[j] = d;
            }
            /* outputs k = (2i+1) << level, i < half: dot(o, matrix row k) */
            for (int i = 0; i < half; i++) {
                const int k = (2 * i + 1) << level;
                int32_t acc = 0;
                for (int j = 0; j < half; j++)
                    acc += tcoef(n, k, j) * o[j];
                out[k * dstStride + r] = (int16_t)((acc + offset) >> shift);
            }
            for (int j = 0; j < half; j++)
                e[j] = ne[j];
            len = half;
            level++;
        }
        /* len == 2: outputs 0 and n/2 */
        out[0 * dstStride + r] = (int16_t)((tcoef(n, 0, 0) * e[0] + tcoef(n, 0, 1) * e[1] + offset) >> shift);
        out[(n / 2) * dstStride + r] = (int16_t)((tcoef(n, n / 2, 0) * e[0] + tcoef(n, n / 2, 1) * e[1] + offset) >> shift);
    }
}

static inline int16_t clip16(int32_t v) { return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

/* 1-D inverse (PartialButterflyInverse*, EbTransforms_C.c:1075-1556): exact matrix product */
static void inv_1d(const int16_t *in, int srcStride, int16_t *out, int dstStride, int n, int shift)
{
    const int16_t offset = (int16_t)(1 << (shift - 1));
    for (int r = 0; r < n; r++)
        for (int j = 0; j < n; j++) {
            int32_t acc = 0;
            for (int k = 0; k < n; k++)
                acc += tcoef(n, k, j) * in[k * srcStride + r];
            out[r * dstStride + j] = clip16((acc + offset) >> shift);
        }
}

/* Dst4 / DstInverse4, EbTransforms_C.c:1033-1073, 1558-1600 */
static void dst4_fwd(const int16_t *in, int srcStride, int16_t *out, int dstStride, int shift)
{
    const int16_t offset = (int16_t)(1 << (shift - 1));
    for (int r = 0; r < 4; r++) {
        const int16_t *x = in + r * srcStride;
        const int32_t e0 = x[0] + x[3], o0 = x[1] + x[3], e1 = x[0] - x[1], o1 = 74 * x[2];
        out[0 * dstStride + r] = (int16_t)((29 * e0 + 55 * o0 + o1 + offset) >> shift);
        out[2 * dstStride + r] = (int16_t)((29 * e1 + 55 * e0 - o1 + offset) >> shift);
        out[1 * dstStride + r] = (int16_t)(((74 * (x[0] + x[1] - x[3])) + offset) >> shift);
        out[3 * dstStride + r] = (int16_t)((55 * e1 - 29 * o0 + o1 + offset) >> shift);
    }
}
static void dst4_inv(const int16_t *in, int srcStride, int16_t *out, int dstStride, int shift)
{
    const int16_t offset = (int16_t)(1 << (shift - 1));
    for (int r = 0; r < 4; r++) {
        const int32_t c0 = in[r], c1 = in[srcStride + r], c2 = in[2 * srcStride + r], c3 = in[3 * srcStride + r];
        const int32_t o0 = c0 + c2, o1 = c0 - c3, e0 = c2 + c3, e1 = 74 * c1;
        out[r * dstStride + 0] = clip16((29 * o0 + 55 * e0 + e1 + offset) >> shift);
        out[r * dstStride + 1] = clip16((55 * o1 - 29 * e0 + e1 + offset) >> shift);
        out[r * dstStride + 2] = clip16(((74 * (c0 - c2 + c3)) + offset) >> shift);
        out[r * dstStride + 3] = clip16((55 * o0 + 29 * o1 - e1 + offset) >> shift);
    }
}

/* kind: 0 DCT (full precision), 1 DCT "Estimate" (32/16 only), 2 DST (4x4 only).
 * Transform32x32/16x16(+Estimate)/8x8/4x4/DstTransform4x4, EbTransforms_C.c:1602-1908:
 * first-pass shifts 4/3/2/1 + bitIncrement (Estimate: 6/4 + bitIncrement),
 * second-pass shifts 11/10/9/8 (Estimate: 9/9). */
void svt_oracle_FwdTransform(int kind, int size, const int16_t *residual, uint32_t srcStride, int16_t *coeff,
                             uint32_t dstStride, int16_t *inner, uint32_t bitIncrement)
{
    int16_t tmp[32 * 32];
    if (!inner)
        inner = tmp;
    const int lg = size == 32 ? 5 : size == 16 ? 4 : size == 8 ? 3 : 2;
    if (kind == 2) {
        dst4_fwd(residual, (int)srcStride, inner, 4, 1 + (int)bitIncrement);
        dst4_fwd(inner, 4, coeff, (int)dstStride, 8);
        return;
    }
    const int est = (kind == 1 && size >= 16);
    const int s1 = est ? (size == 32 ? 6 : 4) + (int)bitIncrement : lg - 1 + (int)bitIncrement;
    const int s2 = est ? 9 : lg + 6;
    const int wrap = est ? (size == 32 ? 2 : 1) : 0;
    fwd_1d(residual, (int)srcStride, inner, size, size, s1, wrap);
    fwd_1d(inner, size, coeff, (int)dstStride, size, s2, wrap);
}

/* InvTransform32x32/16x16/8x8/4x4/InvDstTransform4x4, EbTransforms_C.c:1910-2119:
 * shifts SHIFT_INV_1ST = 7, SHIFT_INV_2ND - bitIncrement = 12 - bitIncrement. */
void svt_oracle_InvTransform(int kind, int size, const int16_t *coeff, uint32_t srcStride, int16_t *residual,
                             uint32_t dstStride, int16_t *inner, uint32_t bitIncrement)
{
    int16_t tmp[32 * 32];
    if (!inner)
        inner = tmp;
    if (kind == 2) {
        dst4_inv(coeff, (int)srcStride, inner, 4, 7);
        dst4_inv(inner, 4, residual, (int)dstStride, 12 - (int)bitIncrement);
        return;
    }
    inv_1d(coeff, (int)srcStride, inner, size, size, 7);
    inv_1d(inner, size, residual, (int)dstStride, size, 12 - (int)bitIncrement);
}

/* QuantizeInvQuantize, EbTransforms_C.c:89-138 */
void svt_oracle_QuantizeInvQuantize(const int16_t *coeff, uint32_t coeffStride, int16_t *quantCoeff,
                                    int16_t *reconCoeff, uint32_t qFunc, uint32_t q_offset, int32_t shiftedQBits,
                                    int32_t shiftedFFunc, int32_t iq_offset, int32_t shiftNum, uint32_t areaSize,
                                    uint32_t *nonzerocoeff)
{
    uint32_t nz = 0;
    for (uint32_t r = 0; r < areaSize; r++)
        for (uint32_t c = 0; c < areaSize; c++) {
            const uint32_t loc = r * coeffStride + c;
            const int32_t v = coeff[loc], sign = v < 0 ? -1 : 1;
            int32_t t = abs(v);
            t = (int32_t)((uint32_t)t * qFunc); /* int *= unsigned: converted to unsigned, back to int */
            t = (int32_t)((uint32_t)t + q_offset);
            t >>= shiftedQBits;
            const int16_t q = clip16(sign * t);
            quantCoeff[loc] = q;
            nz += q != 0;
            reconCoeff[loc] = clip16(((q * shiftedFFunc) + iq_offset) >> shiftNum);
        }
    *nonzerocoeff = nz;
}

/* UpdateQiQCoef, EbTransforms_C.c:209-260 (sliceType: 2 == EB_I_PICTURE) */
void svt_oracle_UpdateQiQCoef(int16_t *quantCoeff, int16_t *reconCoeff, uint32_t coeffStride, int32_t shiftedFFunc,
                              int32_t iq_offset, int32_t shiftNum, uint32_t areaSize, uint32_t *nonzerocoeff,
                              uint32_t componentType, uint32_t sliceType, uint32_t temporalLayer,
                              uint32_t enableCbflag, uint8_t enableContouringQCUpdateFlag)
{
    if ((*nonzerocoeff < 10) && enableContouringQCUpdateFlag && sliceType == 2 && temporalLayer == 0 && componentType == 0) {
        const uint32_t loc = (areaSize - 1) + (areaSize - 1) * coeffStride;
        if (quantCoeff[loc] == 0) {
            (*nonzerocoeff)++;
            quantCoeff[loc] = 1;
            reconCoeff[loc] = (int16_t)((int16_t)((quantCoeff[loc] * shiftedFFunc) + iq_offset) >> shiftNum);
        }
    }
    if ((*nonzerocoeff == 0) && (enableCbflag == 1)) {
        const uint32_t loc = ((areaSize - 2) * coeffStride) + (areaSize - 1);
        *nonzerocoeff = 1;
        quantCoeff[loc] = 1;
        reconCoeff[loc] = (int16_t)((int16_t)((quantCoeff[loc] * shiftedFFunc) + iq_offset) >> shiftNum);
    }
}

/* ResidualKernel / PictureAdditionKernel / ZeroOutCoeffKernel, C_DEFAULT/EbPictureOperators_C.c:112-360 */
void svt_oracle_ResidualKernel(const uint8_t *input, uint32_t inputStride, const uint8_t *pred, uint32_t predStride,
                               int16_t *residual, uint32_t residualStride, uint32_t w, uint32_t h)
{
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++)
            residual[y * residualStride + x] = (int16_t)((int16_t)input[y * inputStride + x] - (int16_t)pred[y * predStride + x]);
}
void svt_oracle_PictureAdditionKernel(const uint8_t *pred, uint32_t predStride, const int16_t *residual,
                                      uint32_t residualStride, uint8_t *recon, uint32_t reconStride, uint32_t w,
                                      uint32_t h)
{
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const int32_t v = (int32_t)residual[y * residualStride + x] + pred[y * predStride + x];
            recon[y * reconStride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}
void svt_oracle_ZeroOutCoeffKernel(int16_t *coeff, uint32_t stride, uint32_t origin, uint32_t w, uint32_t h)
{
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++)
            coeff[y * stride + x + origin] = 0;
}

/* FullDistortionKernel_32bit / CbfZero / Intra, EbPictureOperators_C.c:385-480: the difference
 * is truncated to int16 before squaring (SQR16to32 takes EB_S16) and the sums are 32-bit. */
void svt_oracle_FullDistortionKernel_32bit(const int16_t *coeff, uint32_t coeffStride, const int16_t *recon,
                                           uint32_t reconStride, uint64_t result[2], uint32_t w, uint32_t h, int mode)
{
    uint32_t res = 0, pred = 0;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const int16_t d = (int16_t)(coeff[y * coeffStride + x] - recon[y * reconStride + x]);
            const int16_t c = coeff[y * coeffStride + x];
            res += (uint32_t)(d * d);
            pred += (uint32_t)(c * c);
        }
    if (mode == 0) /* FullDistortionKernel_32bit */
        result[0] = res, result[1] = pred;
    else if (mode == 1) /* FullDistortionKernelCbfZero_32bit */
        result[0] = pred, result[1] = pred;
    else /* FullDistortionKernelIntra_32bit */
        result[0] = res, result[1] = res;
}

/* Compute8x8Satd / Compute8x8Satd_U8 (EbPictureOperators_C.c:481-642) and Compute4x4Satd /
 * Compute4x4Satd_U8 (Codec/EbHmCode.c:41-215): 2-D Hadamard in 16-bit arithmetic, sum of
 * magnitudes, (s+2)>>2 resp. (s+1)>>1; the _U8 forms also accumulate the DC term. */
static uint64_t hadamard_sum(const int16_t *d, int n, int16_t *dc)
{
    int16_t m[64];
    for (int i = 0; i < n * n; i++)
        m[i] = d[i];
    for (int pass = 0; pass < 2; pass++) /* rows then columns */
        for (int line = 0; line < n; line++)
            for (int len = 1; len < n; len <<= 1)
                for (int i = 0; i < n; i += len << 1)
                    for (int j = i; j < i + len; j++) {
                        int16_t *a = pass ? &m[j * n + line] : &m[line * n + j];
                        int16_t *b = pass ? &m[(j + len) * n + line] : &m[line * n + j + len];
                        const int16_t s = (int16_t)(*a + *b), t = (int16_t)(*a - *b);
                        *a = s, *b = t;
                    }
    uint64_t sum = 0;
    for (int i = 0; i < n * n; i++)
        sum += (uint64_t)abs(m[i]);
    if (dc)
        *dc = m[0];
    return sum;
}
uint64_t svt_oracle_Compute8x8Satd(const int16_t *diff) { return (hadamard_sum(diff, 8, NULL) + 2) >> 2; }
uint64_t svt_oracle_Compute4x4Satd(const int16_t *diff) { return (hadamard_sum(diff, 4, NULL) + 1) >> 1; }
uint64_t svt_oracle_Compute8x8Satd_U8(const uint8_t *src, uint64_t *dcValue, uint32_t srcStride)
{
    int16_t d[64], dc;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++)
            d[y * 8 + x] = src[y * srcStride + x];
    const uint64_t s = (hadamard_sum(d, 8, &dc) + 2) >> 2;
    *dcValue += (uint64_t)(int64_t)dc;
    return s;
}
uint64_t svt_oracle_Compute4x4Satd_U8(const uint8_t *src, uint64_t *dcValue, uint32_t srcStride)
{
    int16_t d[16], dc;
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++)
            d[y * 4 + x] = src[y * srcStride + x];
    const uint64_t s = (hadamard_sum(d, 4, &dc) + 1) >> 1;
    *dcValue += (uint64_t)(int64_t)dc;
    return s;
}

/* ------------------------------------------------------------------------------------------------------------------
 * UnifiedQuantizeInvQuantize (Codec/EbTransforms.c:2978-3250) on its paths without RDOQ / PM-core and without perceptual
 * masking.  coeff / quant / recon: size x size blocks with row pitch `stride`.  Pinned by
 * tests/test_oracle_uqiq_golden.py on records of real encode-pass calls.
 * ------------------------------------------------------------------------------------------------------------------ */
void svt_oracle_unified_quantize(const SvtAmdQuantUnit *U, const int16_t *coeff, uint32_t stride, int16_t *quant, int16_t *recon,
                                 uint32_t *nzOut)
{
    static const uint32_t QF[6] = {26214, 23302, 20560, 18396, 16384, 14564}, FF[6] = {40, 45, 51, 57, 64, 72};
    uint32_t lg = 0;
    while ((1u << lg) < U->size)
        lg++;
    const int32_t qpRem = U->qp % 6, qpPer = U->qp / 6;
    const uint32_t tshift = 15 - U->bit_depth - lg;
    const int32_t shiftedQBits = 14 + qpPer + (int32_t)tshift;
    const uint32_t q_offset = ((U->slice_type == 2 || U->slice_type == 3) ? 171u : 85u) << (shiftedQBits - 9);
    const int32_t shiftedFFunc = qpPer > 8 ? (int32_t)FF[qpRem] << (qpPer - 2) : (int32_t)FF[qpRem] << qpPer;
    const int32_t shiftNum = qpPer > 8 ? 20 - 14 - (int32_t)tshift - 2 : 20 - 14 - (int32_t)tshift;
    const int32_t iq_offset = 1 << (shiftNum - 1);
    uint32_t nz = 0;
    if (U->shape == 3) { /* ONLY_DC_SHAPE (:3043-3090) */
        const int32_t c = coeff[0], sign = c < 0 ? -1 : 1;
        int32_t t = (c < 0 ? -c : c) * (int32_t)QF[qpRem];
        t = (int32_t)((uint32_t)t + q_offset);
        t >>= shiftedQBits;
        int32_t q = sign * t;
        q = q < -32768 ? -32768 : q > 32767 ? 32767 : q;
        quant[0] = (int16_t)q;
        nz = q != 0;
        int32_t r = ((q * shiftedFFunc) + iq_offset) >> shiftNum;
        recon[0] = (int16_t)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
        *nzOut = nz;
        return;
    }
    const uint32_t offs = U->dz_offset ? (uint32_t)(U->dz_offset * (1u << shiftedQBits) / 20) : q_offset;
    const uint32_t area = (uint32_t)U->size >> U->shape;
    svt_oracle_QuantizeInvQuantize(coeff, stride, quant, recon, QF[qpRem], offs, shiftedQBits, shiftedFFunc, iq_offset, shiftNum, area,
                                   &nz);
    if (U->clean_sparse && U->size >= 8 && nz && U->slice_type != 2 && area >= 4) /* :3190-3232 */
        for (uint32_t by = 0; by < area / 4; by++)
            for (uint32_t bx = 0; bx < area / 4; bx++) {
                uint32_t cnt = 0;
                for (uint32_t y = 0; y < 4; y++)
                    for (uint32_t x = 0; x < 4; x++)
                        cnt += quant[(4 * by + y) * stride + 4 * bx + x] != 0;
                if (cnt == 1)
                    for (uint32_t y = 0; y < 4; y++)
                        for (uint32_t x = 0; x < 4; x++) {
                            const uint32_t loc = (4 * by + y) * stride + 4 * bx + x;
                            if (quant[loc] && loc)
                                quant[loc] = 0, recon[loc] = 0, nz--;
                        }
            }
    svt_oracle_UpdateQiQCoef(quant, recon, stride, shiftedFFunc, iq_offset, shiftNum, area, &nz, U->component, U->slice_type,
                             U->temporal_layer, U->enable_cb_flag, U->contouring_flag);
    *nzOut = nz;
}
