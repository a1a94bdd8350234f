/*
 * oracle/svt_oracle_sbo.c - TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 * CPU restatement of the source-based-operations input of the mode-decision configuration:
 *   ComputeNxMSatdSadLCU     Codec/EbPictureOperators.c:232-263   (blocks of at least 8x8: EbHevcComputeNxMSatd8x8Units_U8 :186)
 *   Compute8x8Satd_U8        C_DEFAULT/EbPictureOperators_C.c:563-641
 *   CalculateAcEnergy        Codec/EbSourceBasedOperationsProcess.c:302-362 (the 64x64 and its four 32x32 of a complete LCU; 100000000 otherwise)
 * Pinned by tests/test_oracle_sbo.py on the reference's own symbol (oracle/_ref/libsvtref.so: ComputeNxMSatdSadLCU, the dispatch the encoder runs).
 */
#include <stdlib.h>
#include "svt_oracle.h"

static uint64_t satd8x8(const uint8_t *src, uint32_t stride, uint64_t *dc)
{
    int32_t a[8][8], t[8];
    for (int j = 0; j < 8; j++) { /* rows: three butterfly stages, distance 4, 2, 1 */
        for (int i = 0; i < 8; i++)
            a[j][i] = src[j * stride + i];
        for (int span = 4; span; span >>= 1) {
            for (int i = 0; i < 8; i++)
                t[i] = (i & span) ? a[j][i - span] - a[j][i] : a[j][i] + a[j][i + span];
            for (int i = 0; i < 8; i++)
                a[j][i] = t[i];
        }
    }
    for (int i = 0; i < 8; i++) /* columns */
        for (int span = 4; span; span >>= 1) {
            for (int j = 0; j < 8; j++)
                t[j] = (j & span) ? a[j - span][i] - a[j][i] : a[j][i] + a[j + span][i];
            for (int j = 0; j < 8; j++)
                a[j][i] = t[j];
        }
    uint64_t s = 0;
    for (int j = 0; j < 8; j++)
        for (int i = 0; i < 8; i++)
            s += (uint64_t)abs(a[j][i]);
    *dc += (uint64_t)a[0][0];
    return (s + 2) >> 2;
}

/* width, height: multiples of 8 (the only sizes CalculateAcEnergy asks for are 64 and 32) */
uint64_t svt_oracle_sbo_ac_energy(const uint8_t *src, uint32_t stride, uint32_t width, uint32_t height)
{
    uint64_t satd = 0, dc = 0;
    for (uint32_t y = 0; y < height; y += 8)
        for (uint32_t x = 0; x < width; x += 8)
            satd += satd8x8(src + (size_t)y * stride + x, stride, &dc);
    return satd - (dc >> 2);
}

/* luma: sample (0,0) of the picture; out[lcu][5] */
void svt_oracle_sbo_ac_energy_picture(const uint8_t *luma, uint32_t stride, uint32_t width, uint32_t height, uint64_t *out)
{
    const uint32_t wl = (width + 63) / 64, hl = (height + 63) / 64;
    for (uint32_t k = 0; k < wl * hl; k++) {
        const uint32_t x = 64 * (k % wl), y = 64 * (k / wl);
        if (x + 64 > width || y + 64 > height) {
            for (int i = 0; i < 5; i++)
                out[k * 5 + i] = 100000000ull;
            continue;
        }
        out[k * 5] = svt_oracle_sbo_ac_energy(luma + (size_t)y * stride + x, stride, 64, 64);
        for (int q = 0; q < 4; q++)
            out[k * 5 + 1 + q] = svt_oracle_sbo_ac_energy(luma + (size_t)(y + 32 * (q >> 1)) * stride + x + 32 * (q & 1), stride, 32, 32);
    }
}
