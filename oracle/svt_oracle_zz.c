/*
 * oracle/svt_oracle_zz.c - TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 * CPU restatement of ComputeDecimatedZzSad (Codec/EbMotionEstimationProcess.c:176-300): per complete LCU the 16x16 SAD
 * between the 1/16 current picture and the collocated LCU of the previous input picture decimated by 4
 * (Decimation2D :239, NxMSadKernel [2] :249), then the two background-classification ladders
 * (BEA_CLASS_* thresholds, Codec/EbDefinitions.h:1087-1107).  The function is `static` in the reference, so it is
 * pinned through its two leaf calls (Decimation2D and NxMSadKernel are pinned in tests/test_oracle_leaf.py) and the
 * constants quoted here.
 */
#include "svt_oracle.h"

void svt_oracle_zz_sad_picture(const uint8_t *cur, const uint8_t *prev, uint32_t stride, uint32_t width, uint32_t height,
                               SvtAmdZzLcu *out)
{
    const uint32_t wl = (width + 63) / 64, hl = (height + 63) / 64;
    for (uint32_t ly = 0; ly < hl; ly++)
        for (uint32_t lx = 0; lx < wl; lx++) {
            SvtAmdZzLcu *o = &out[ly * wl + lx];
            const uint32_t ox = lx * 64, oy = ly * 64;
            const uint32_t lw = width - ox < 64 ? width - ox : 64, lh = height - oy < 64 ? height - oy : 64;
            uint32_t sad;
            if (lw == 64 && lh == 64) { /* lcuParams->isCompleteLcu */
                uint8_t c16[16 * 16], p16[16 * 16];
                svt_oracle_Decimation2D(cur + (size_t)oy * stride + ox, stride, 64, 64, c16, 16, 4);  /* = the 1/16 plane */
                svt_oracle_Decimation2D(prev + (size_t)oy * stride + ox, stride, 64, 64, p16, 16, 4);
                sad = svt_oracle_NxMSadKernel(c16, 16, p16, 16, 16, 16);
                o->zz_cost = sad < 16 * 16 ? 0 : sad < 16 * 16 * 2 ? 3 : sad < 16 * 16 * 4 ? 10 : sad < 16 * 16 * 8 ? 20 : 30;
            } else {
                sad = ~0u;
                o->zz_cost = 0xFF; /* INVALID_ZZ_COST */
            }
            const uint32_t area = (lw >> 2) * (lh >> 2);
            o->non_moving_index = sad < area * 2 ? 0 : sad < area * 4 ? 10 : sad < area * 8 ? 20 : 30;
            o->sad = sad;
            o->pad[0] = o->pad[1] = 0;
        }
}
