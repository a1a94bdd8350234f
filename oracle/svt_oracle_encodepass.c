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
 *
 * Inter 2Nx2N units (EbCodingLoop.c:3817-4400; the host has made the merge / skip decision, SvtAmdLcuCu.inter_kind) add, from pinned pieces:
 *     svt_oracle_inter_pu[16bit]            EncodePassInterPrediction (tests/test_oracle_inter_golden.py)
 *     svt_oracle_FullDistortionKernel_32bit PictureFullDistortionLuma (tests/test_oracle_leaf.py)
 *     svt_oracle_coeff_bits_lossy           TuEstimateCoeffBitsEncDec -> EstimateQuantizedCoefficients[1] (tests/test_oracle_rate.py)
 * and the luma cbf decision EncodeTuCalcCost (EbRateDistortionCost.c:2578-2675) restated below; the glue is pinned on recorded EncodePass
 * calls of P / B pictures (tests/golden/encodepass_p_*.npz, _b_*).
 */
#include <stdio.h>
#include <stdlib.h>
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
/* the encode pass's quantiser of one unit: UnifiedQuantizeInvQuantize, or with SvtAmdLcuWork.pm_core its EB_PMCORE branch
 * (DecoupledQuantizeInvQuantizeLoops, Codec/EbTransforms.c:3009-3052; svt_oracle_pmcore_quantize, pinned by tests/test_oracle_uqiq_golden.py) */
static void ep_quantize(int bps, const SvtAmdLcuWork *W, const SvtAmdLcuCu *cu, int p, int n, const SvtAmdCabacCost *cost, const int16_t *coeff, int16_t *q,
                        int16_t *r, uint32_t *nz)
{
    const uint8_t qp = (uint8_t)((p ? cu->chroma_qp : cu->qp) + (bps == 2 ? 12 : 0));
    if (W->pm_core) {
        SvtAmdPmQuantUnit U;
        memset(&U, 0, sizeof(U));
        U.size = (uint8_t)n, U.qp = qp, U.bit_depth = bps == 1 ? 8 : 10, U.slice_type = W->slice_type, U.component = p ? 1 : 0;
        U.cand_type = cu->pred_mode, U.lambda = W->full_lambda;
        svt_oracle_pmcore_quantize(cost, &U, coeff, q, r, nz);
        return;
    }
    SvtAmdQuantUnit U;
    memset(&U, 0, sizeof(U));
    U.size = (uint8_t)n, U.qp = qp, U.bit_depth = bps == 1 ? 8 : 10;
    U.slice_type = W->slice_type, U.component = p ? 1 : 0, U.temporal_layer = W->temporal_layer, U.dz_offset = p ? 0 : cu->dz_offset;
    svt_oracle_unified_quantize(&U, coeff, (uint32_t)n, q, r, nz);
}

static int ilog2i(int v)
{
    int l = 0;
    while ((1 << l) < v)
        l++;
    return l;
}

/* one inter unit: prediction, then per transform unit EncodeLoop (+ the luma cbf decision of AMVP units) and EncodeGenerateRecon */
static void encode_inter_cu(int bps, void *const rec[3], const uint32_t pitch[3], const SvtAmdLcuWork *W, const SvtAmdLcuCu *cu, const void *const srcp[3],
                            SvtAmdLcuCuResult *Rcu, int16_t *const coeffp[3], const SvtAmdRefPicture *ref0, const SvtAmdRefPicture *ref1,
                            const SvtAmdCabacCost *cost)
{
    const int N = cu->size, x0 = W->lcu_x + cu->x, y0 = W->lcu_y + cu->y;
    uint8_t *d[3];
    for (int p = 0; p < 3; p++)
        d[p] = (uint8_t *)rec[p] + ((size_t)(p ? y0 >> 1 : y0) * pitch[p] + (p ? x0 >> 1 : x0)) * (size_t)bps;
    SvtAmdInterPuJob J;
    memset(&J, 0, sizeof(J));
    memcpy(J.mv, cu->mv, sizeof(J.mv));
    J.pu_x = (uint16_t)x0, J.pu_y = (uint16_t)y0, J.pu_w = J.pu_h = (uint8_t)N, J.pred_dir = cu->inter_dir;
    if (bps == 1)
        svt_oracle_inter_pu(&J, ref0, ref1, d[0], pitch[0], d[1], d[2], pitch[1]);
    else
        svt_oracle_inter_pu16bit(&J, ref0, ref1, (uint16_t *)d[0], pitch[0], (uint16_t *)d[1], (uint16_t *)d[2], pitch[1]);
    const int ntu = N == 64 ? 4 : 1, T = N == 64 ? 32 : N;
    SvtAmdLcuCuResult *agg = Rcu; /* transformUnitArray[0] */
    memset(agg, 0, sizeof(*agg));
    for (int tu = 0; tu < ntu; tu++) {
        SvtAmdLcuCuResult *o = N == 64 ? Rcu + 1 + tu : Rcu;
        const int tx = cu->x + ((tu & 1) << 5), ty = cu->y + ((tu > 1) << 5); /* inside the LCU */
        memset(o, 0, sizeof(*o));
        if (cu->inter_kind == SVT_AMD_EP_INTER_SKIP)
            continue; /* :4165-4171 */
        for (int p = 0; p < 3; p++) {
            const int n = p ? T >> 1 : T, lx = p ? tx >> 1 : tx, ly = p ? ty >> 1 : ty, sp = p ? 32 : 64;
            uint8_t *dp = (uint8_t *)rec[p] + ((size_t)((p ? W->lcu_y >> 1 : W->lcu_y) + ly) * pitch[p] + (p ? W->lcu_x >> 1 : W->lcu_x) + lx) * (size_t)bps;
            int16_t res[32 * 32], coeff[32 * 32], q[32 * 32], r[32 * 32];
            for (int j = 0; j < n; j++)
                for (int i = 0; i < n; i++)
                    res[j * n + i] = (int16_t)(rd(srcp[p], (size_t)(ly + j) * sp + lx + i, bps) - rd(dp, (size_t)j * pitch[p] + i, bps));
            svt_oracle_FwdTransform(n >= 16 ? 1 : 0, n, res, (uint32_t)n, coeff, (uint32_t)n, NULL, bps == 1 ? 0 : 2);
            uint32_t nz = 0;
            ep_quantize(bps, W, cu, p, n, cost, coeff, q, r, &nz);
            const int only_dc = nz == 1 && r[0] != 0 && !(p == 0 && n == 32);
            int cbf = nz != 0;
            if (p == 0 && cu->inter_kind == SVT_AMD_EP_INTER_AMVP) {
                /* PictureFullDistortionLuma over the unit (its DC alone for a DC-only unit) -> TuEstimateCoeffBitsEncDec -> EncodeTuCalcCost
                 * (EbCodingLoop.c:4075-4124) */
                uint64_t dd[2] = {0, 0};
                const uint32_t area = only_dc ? 1u : (uint32_t)n;
                svt_oracle_FullDistortionKernel_32bit(coeff, (uint32_t)n, r, (uint32_t)n, dd, area, area, nz == 0 ? 1 : 0);
                const int shift = 2 * (7 - ilog2i(n));
                dd[0] = (dd[0] + ((uint64_t)1 << (shift - 1))) >> shift, dd[1] = (dd[1] + ((uint64_t)1 << (shift - 1))) >> shift;
                uint64_t bits = nz ? svt_oracle_coeff_bits_lossy(cost, (uint32_t)n, 1 /* INTER_MODE */, 0xFF, 0xFF, q, (uint32_t)n, 0, nz) : 0;
                bits >>= 15;
                const uint32_t ctx = N == T;
                const uint64_t nzRate = (bits << 15) + W->luma_cbf_bits[2 + ctx], zRate = W->luma_cbf_bits[ctx];
                const uint64_t zCost = (dd[1] << 8) + ((((uint64_t)W->full_lambda * zRate) + (1u << 22)) >> 23);
                const uint64_t nzCost = (dd[0] << 8) + ((((uint64_t)W->full_lambda * nzRate) + (1u << 22)) >> 23);
                cbf = nz != 0 && nzCost < zCost;
                if (getenv("SVT_ORACLE_EP_DEBUG"))
                    fprintf(stderr, "oracle AMVP unit (%d,%d) n %d nz %u only_dc %d d0 %llu d1 %llu bits %llu lambda %u cbfbits %u %u zCost %llu nzCost %llu -> cbf %d\n",
                            cu->x, cu->y, n, nz, only_dc, (unsigned long long)dd[0], (unsigned long long)dd[1], (unsigned long long)bits, W->full_lambda,
                            W->luma_cbf_bits[ctx], W->luma_cbf_bits[2 + ctx], (unsigned long long)zCost, (unsigned long long)nzCost, cbf);
            }
            if (cbf)
                svt_oracle_recon_tu(bps, (uint32_t)n, only_dc, 0, r, dp, pitch[p], dp, pitch[p]);
            int16_t *cq = coeffp[p] + ly * sp + lx;
            for (int j = 0; j < n; j++)
                memcpy(cq + j * sp, q + j * n, (size_t)n * 2);
            o->cbf[p] = (uint8_t)cbf, o->only_dc[p] = (uint8_t)only_dc, o->nz[p] = (uint16_t)nz;
            /* transformUnitArray[0] collects the flags: chroma always (:4263-4281), luma only where EncodeTuCalcCost runs (its tail) */
            if (N == 64 && cbf && (p != 0 || cu->inter_kind == SVT_AMD_EP_INTER_AMVP))
                agg->cbf[p] = 1;
        }
    }
}

static void encode_lcu(int bps, void *const rec[3], const uint32_t pitch[3], uint8_t *map, uint32_t mapPitch, uint32_t width, uint32_t height,
                       const SvtAmdLcuWork *W, const void *const srcp[3], SvtAmdLcuCuResult *Rcu, int16_t *const coeffp[3], void *const recout[3],
                       const SvtAmdRefPicture *ref0, const SvtAmdRefPicture *ref1, const SvtAmdCabacCost *cost)
{
    if (W->pm_core && !cost)
        return;
    for (int ci = 0; ci < W->num_cus; ci++) {
        const SvtAmdLcuCu *cu = &W->cu[ci];
        const int N = cu->size, x0 = W->lcu_x + cu->x, y0 = W->lcu_y + cu->y;
        if (cu->pred_mode == 1) { /* INTER_MODE */
            if (pitch[1] != pitch[2] || (!ref0 && !ref1) || !cost)
                return;
            encode_inter_cu(bps, rec, pitch, W, cu, srcp, &Rcu[ci], coeffp, ref0, ref1, cost);
            for (int j = 0; j < N / 4; j++)
                memset(map + (size_t)((y0 >> 2) + j) * mapPitch + (x0 >> 2), cu->pred_mode, (size_t)N / 4);
            continue;
        }
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
            uint32_t nz = 0;
            ep_quantize(bps, W, cu, p, n, cost, coeff, q, r, &nz);
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
    encode_lcu(1, rp, pitch, map, mapPitch, width, height, W, sp, R->cu, cp, ro, NULL, NULL, NULL);
}

void svt_oracle_encode_lcu16(uint16_t *const rec[3], const uint32_t pitch[3], uint8_t *map, uint32_t mapPitch, uint32_t width, uint32_t height,
                             const SvtAmdLcuWork16 *W, SvtAmdLcuResult16 *R)
{
    memset(R, 0, sizeof(*R));
    void *const rp[3] = {rec[0], rec[1], rec[2]};
    const void *const sp[3] = {W->src_y, W->src_cb, W->src_cr};
    int16_t *const cp[3] = {R->coeff_y, R->coeff_cb, R->coeff_cr};
    void *const ro[3] = {R->rec_y, R->rec_cb, R->rec_cr};
    encode_lcu(2, rp, pitch, map, mapPitch, width, height, (const SvtAmdLcuWork *)W /* same head */, sp, R->cu, cp, ro, NULL, NULL, NULL);
}

/* P / B pictures: reference pictures of list 0 / 1 (host pointers in the d_* fields, as svt_oracle_inter_pu takes them) and the picture's
 * coefficient-rate tables */
void svt_oracle_encode_lcu_inter(uint8_t *const rec[3], const uint32_t pitch[3], uint8_t *map, uint32_t mapPitch, uint32_t width, uint32_t height,
                                 const SvtAmdRefPicture *ref0, const SvtAmdRefPicture *ref1, const SvtAmdCabacCost *cost, const SvtAmdLcuWork *W,
                                 SvtAmdLcuResult *R)
{
    memset(R, 0, sizeof(*R));
    void *const rp[3] = {rec[0], rec[1], rec[2]};
    const void *const sp[3] = {W->src_y, W->src_cb, W->src_cr};
    int16_t *const cp[3] = {R->coeff_y, R->coeff_cb, R->coeff_cr};
    void *const ro[3] = {R->rec_y, R->rec_cb, R->rec_cr};
    encode_lcu(1, rp, pitch, map, mapPitch, width, height, W, sp, R->cu, cp, ro, ref0, ref1, cost);
}

void svt_oracle_encode_lcu_inter16(uint16_t *const rec[3], const uint32_t pitch[3], uint8_t *map, uint32_t mapPitch, uint32_t width, uint32_t height,
                                   const SvtAmdRefPicture *ref0, const SvtAmdRefPicture *ref1, const SvtAmdCabacCost *cost, const SvtAmdLcuWork16 *W,
                                   SvtAmdLcuResult16 *R)
{
    memset(R, 0, sizeof(*R));
    void *const rp[3] = {rec[0], rec[1], rec[2]};
    const void *const sp[3] = {W->src_y, W->src_cb, W->src_cr};
    int16_t *const cp[3] = {R->coeff_y, R->coeff_cb, R->coeff_cr};
    void *const ro[3] = {R->rec_y, R->rec_cb, R->rec_cr};
    encode_lcu(2, rp, pitch, map, mapPitch, width, height, (const SvtAmdLcuWork *)W /* same head */, sp, R->cu, cp, ro, ref0, ref1, cost);
}
