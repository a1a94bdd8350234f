/*
 * oracle/svt_oracle_encodepass.c - TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 * CPU restatement of the coding-unit loop of EncodePass (Codec/EbCodingLoop.c:2989, :3180-4594) for an LCU whose units are intra
 * 2Nx2N units of 8..32, composed of restatements that are each pinned on reference data:
 *     svt_oracle_intra_pu          GenerateIntraReferenceSamplesEncodePass + EncodePassIntraPrediction (tests/test_oracle_intra_golden.py)
 *     svt_oracle_FwdTransform      EstimateTransform (tests/test_oracle_txfm.py)
 *     svt_oracle_unified_quantize  UnifiedQuantizeInvQuantize (tests/test_oracle_uqiq_golden.py)
 *     svt_oracle_recon_tu          EncodeGenerateRecon incl. the DC-only shortcut (tests/test_oracle_recon_golden.py)
 * What is new here - and what tests/test_oracle_encodepass_golden.py pins on recorded EncodePass calls - is the glue: the neighbours of a
 * unit are read from the UN-DEBLOCKED reconstruction picture and a per-4x4 mode map instead of the reference's neighbour arrays
 * (EbNeighborArrays.c:113: last row / column of every unit).  A unit's left / top / top-left neighbours lie on the bottom row or right
 * column of the unit that holds them, and that unit is the last writer of the array entry (Z order is monotone in x and y), so both
 * views agree wherever the reference may read.
 */
#include <string.h>
#include "svt_oracle.h"

static int mode_at(const uint8_t *map, uint32_t mapPitch, int w, int h, int px, int py)
{
    if (px < 0 || py < 0 || px >= w || py >= h)
        return 0xFE;
    return map[(size_t)(py >> 2) * mapPitch + (px >> 2)];
}

static int rd(const void *p, size_t i, int bps) { return bps == 1 ? ((const uint8_t *)p)[i] : ((const uint16_t *)p)[i]; }

/* rec[3]: un-deblocked reconstruction planes of the picture (pitch in samples, bps bytes per sample), updated in place; map: one byte
 * per 4x4 luma block (0xFF = not coded yet), updated in place.  bps 1: W / R are SvtAmdLcuWork / SvtAmdLcuResult; bps 2 (EncodePass
 * with is16bit, EncodeLoop16bit :1244: 10-bit samples, quantiser at qp + QP_BD_OFFSET :1307): SvtAmdLcuWork16 / SvtAmdLcuResult16. */
static void encode_lcu(int bps, void *const rec[3], const uint32_t pitch[3], uint8_t *map, uint32_t mapPitch, uint32_t width, uint32_t height,
                       const SvtAmdLcuWork *W, const void *const srcp[3], SvtAmdLcuCuResult *Rcu, int16_t *const coeffp[3], void *const recout[3])
{
    for (int ci = 0; ci < W->num_cus; ci++) {
        const SvtAmdLcuCu *cu = &W->cu[ci];
        const int N = cu->size, x0 = W->lcu_x + cu->x, y0 = W->lcu_y + cu->y;
        SvtAmdIntraPuJob J;
        memset(&J, 0, sizeof(J));
        J.size = (uint32_t)N, J.constrained_intra = W->constrained_intra, J.strong_smoothing = W->strong_smoothing;
        J.pic_left = W->tile_left && cu->x == 0, J.pic_top = W->tile_top && cu->y == 0, J.pic_right = W->tile_right && ((cu->x + N) & 63) == 0;
        J.bottom_left_ok = cu->bottom_left_ok, J.top_right_ok = cu->top_right_ok;
        J.luma_mode = cu->intra_luma_mode, J.chroma_mode = 4; /* EB_INTRA_CHROMA_DM */
        for (int k = 0; k < 2 * N / 4; k++) {
            J.mode_left[k] = (uint8_t)mode_at(map, mapPitch, (int)width, (int)height, x0 - 1, y0 + 4 * k);
            J.mode_top[k] = (uint8_t)mode_at(map, mapPitch, (int)width, (int)height, x0 + 4 * k, y0 - 1);
        }
        const int tl = mode_at(map, mapPitch, (int)width, (int)height, x0 - 1, y0 - 1);
        J.mode_tl = (uint8_t)(tl == 0xFE ? 0xFF : tl);
        for (int p = 0; p < 3; p++) {
            const int sh = p ? 1 : 0, n2 = (2 * N) >> sh, xp = x0 >> sh, yp = y0 >> sh;
            for (int i = 0; i < n2; i++) {
                const int k = (i << sh) >> 2;
                const int le = J.mode_left[k], te = J.mode_top[k];
                J.left[p][i] = (uint16_t)((le == 0xFE || le == 0xFF) ? 0 : rd(rec[p], (size_t)(yp + i) * pitch[p] + xp - 1, bps));
                J.top[p][i] = (uint16_t)((te == 0xFE || te == 0xFF) ? 0 : rd(rec[p], (size_t)(yp - 1) * pitch[p] + xp + i, bps));
            }
            J.tl[p] = (uint16_t)((tl == 0xFE || tl == 0xFF) ? 0 : rd(rec[p], (size_t)(yp - 1) * pitch[p] + xp - 1, bps));
        }
        uint8_t *d[3];
        for (int p = 0; p < 3; p++)
            d[p] = (uint8_t *)rec[p] + ((size_t)(p ? y0 >> 1 : y0) * pitch[p] + (p ? x0 >> 1 : x0)) * (size_t)bps;
        if (pitch[1] != pitch[2])
            return;
        svt_oracle_intra_pu(bps, &J, d[0], pitch[0], d[1], d[2], pitch[1]);
        for (int p = 0; p < 3; p++) {
            const int n = p ? N >> 1 : N, lx = p ? cu->x >> 1 : cu->x, ly = p ? cu->y >> 1 : cu->y, sp = p ? 32 : 64;
            int16_t res[32 * 32], coeff[32 * 32], q[32 * 32], r[32 * 32];
            for (int j = 0; j < n; j++)
                for (int i = 0; i < n; i++)
                    res[j * n + i] = (int16_t)(rd(srcp[p], (size_t)(ly + j) * sp + lx + i, bps) - rd(d[p], (size_t)j * pitch[p] + i, bps));
            svt_oracle_FwdTransform(n >= 16 ? 1 : 0, n, res, (uint32_t)n, coeff, (uint32_t)n, NULL, bps == 1 ? 0 : 2);
            SvtAmdQuantUnit U;
            memset(&U, 0, sizeof(U));
            U.size = (uint8_t)n, U.qp = (uint8_t)((p ? cu->chroma_qp : cu->qp) + (bps == 2 ? 12 : 0)), U.bit_depth = bps == 1 ? 8 : 10;
            U.slice_type = W->slice_type, U.component = p ? 1 : 0, U.temporal_layer = W->temporal_layer, U.dz_offset = p ? 0 : cu->dz_offset;
            uint32_t nz = 0;
            svt_oracle_unified_quantize(&U, coeff, (uint32_t)n, q, r, &nz);
            /* tuPtr->isOnlyDc (EbCodingLoop.c:792, 879, 1000) */
            const int only_dc = nz == 1 && r[0] != 0 && !(p == 0 && n == 32);
            if (nz)
                svt_oracle_recon_tu(bps, (uint32_t)n, only_dc, 0, r, d[p], pitch[p], d[p], pitch[p]);
            int16_t *cq = coeffp[p] + ly * sp + lx;
            for (int j = 0; j < n; j++)
                memcpy(cq + j * sp, q + j * n, (size_t)n * 2);
            Rcu[ci].cbf[p] = nz != 0, Rcu[ci].only_dc[p] = (uint8_t)only_dc, Rcu[ci].nz[p] = (uint16_t)nz;
        }
        for (int j = 0; j < N / 4; j++)
            memset(map + (size_t)((y0 >> 2) + j) * mapPitch + (x0 >> 2), cu->pred_mode, (size_t)N / 4);
    }
    const int lw = (int)width - W->lcu_x < 64 ? (int)width - W->lcu_x : 64, lh = (int)height - W->lcu_y < 64 ? (int)height - W->lcu_y : 64;
    for (int p = 0; p < 3; p++) {
        const int sh = p ? 1 : 0, n = 64 >> sh;
        for (int y = 0; y < lh >> sh; y++)
            memcpy((uint8_t *)recout[p] + (size_t)y * n * bps,
                   (const uint8_t *)rec[p] + ((size_t)((W->lcu_y >> sh) + y) * pitch[p] + (W->lcu_x >> sh)) * (size_t)bps, (size_t)(lw >> sh) * bps);
    }
}

void svt_oracle_encode_lcu(uint8_t *const rec[3], const uint32_t pitch[3], uint8_t *map, uint32_t mapPitch, uint32_t width, uint32_t height,
                           const SvtAmdLcuWork *W, SvtAmdLcuResult *R)
{
    memset(R, 0, sizeof(*R));
    void *const rp[3] = {rec[0], rec[1], rec[2]};
    const void *const sp[3] = {W->src_y, W->src_cb, W->src_cr};
    int16_t *const cp[3] = {R->coeff_y, R->coeff_cb, R->coeff_cr};
    void *const ro[3] = {R->rec_y, R->rec_cb, R->rec_cr};
    encode_lcu(1, rp, pitch, map, mapPitch, width, height, W, sp, R->cu, cp, ro);
}

void svt_oracle_encode_lcu16(uint16_t *const rec[3], const uint32_t pitch[3], uint8_t *map, uint32_t mapPitch, uint32_t width, uint32_t height,
                             const SvtAmdLcuWork16 *W, SvtAmdLcuResult16 *R)
{
    memset(R, 0, sizeof(*R));
    void *const rp[3] = {rec[0], rec[1], rec[2]};
    const void *const sp[3] = {W->src_y, W->src_cb, W->src_cr};
    int16_t *const cp[3] = {R->coeff_y, R->coeff_cb, R->coeff_cr};
    void *const ro[3] = {R->rec_y, R->rec_cb, R->rec_cr};
    encode_lcu(2, rp, pitch, map, mapPitch, width, height, (const SvtAmdLcuWork *)W /* same head */, sp, R->cu, cp, ro);
}
