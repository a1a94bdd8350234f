/*
 * integration/svt_coeff_scan_consumer.h - reference-side binding of the entropy hand-off pre-scan (include/svt_hevc_amd.h
 * "Entropy hand-off pre-scan").  Compiles against the reference's headers (EbEntropyCodingUtil.h: the arithmetic coder's primitives, the context
 * model set, the context-index maps); a maintainer includes it in EbEntropyCoding.c and calls svt_coeff_scan_encode() from EncodeCoeff (:4069)
 * where EncodeQuantizedCoefficientsFuncArray[...] is called today - same arguments minus the coefficient pointer, plus the block's record.
 *
 * It is the CABAC half of EncodeQuantizedCoefficients (Codec/EbEntropyCoding.c:1463-1714) driven by what the device left per block - scan,
 * last position, one significance map / sign word / greater-1 word per 4x4 sub-block, absolute levels in coding order - instead of s16 planes:
 * no coefficient is fetched, re-ordered, compared with zero or negated on the host; every bin and every context model it touches is the
 * reference's (oracle/ref_harness_coeffscan.c runs both on one arithmetic coder state; tests/test_oracle_coeffscan.py compares the bytes).
 */
#ifndef SVT_COEFF_SCAN_CONSUMER_H
#define SVT_COEFF_SCAN_CONSUMER_H
#include "EbEntropyCodingUtil.h"
#include "../include/svt_hevc_amd.h"

static inline void svt_coeff_scan_remainder(CabacEncodeContext_t *cabac, EB_U32 level, EB_U32 base, EB_U32 *rice)
{
    RemainingCoeffExponentialGolombCode(cabac, level - base, *rice);
    if (*rice < 4 && level > (3u << *rice))
        (*rice)++;
}

/* size: transform size of the component (4..32); is_chroma: componentType != COMPONENT_LUMA; groups / levels: the LCU's lists */
static void svt_coeff_scan_encode(CabacEncodeContext_t *cabac, EB_U32 size, EB_U32 is_chroma, const SvtAmdCoeffScanTu *tu,
                                  const SvtAmdCoeffScanGroup *groups, const uint16_t *levels)
{
    static const EB_U8 *const sub_block_scan[4] = {sbScan8, sbScan8, sbScan16, sbScan32};
    BacEncContext_t *bac = &cabac->bacEncContext;
    ContextModelEncContext_t *cm = &cabac->contextModelEncContext;
    const EB_U32 lg = Log2f(size);
    const EB_U32 gt1_base = is_chroma * NUMBER_OF_GREATER_ONE_COEFF_LUMA_CONTEXT_MODELS, gt2_base = is_chroma * NUMBER_OF_GREATER_TWO_COEFF_LUMA_CONTEXT_MODELS;
    if (tu->last_scan_set < 0)
        return;
    if (tu->dc_only) { /* :1308: last position (0, 0), one level */
        const EB_U32 off = is_chroma ? NUMBER_OF_LAST_SIG_XY_CONTEXT_MODELS : ((lg - 2) * 3 + ((lg - 1) >> 2));
        const SvtAmdCoeffScanGroup *g = &groups[tu->first_group];
        const EB_U32 level = levels[g->first_level];
        EncodeOneBin(bac, 0, &cm->lastSigXContextModel[off]);
        EncodeOneBin(bac, 0, &cm->lastSigYContextModel[off]);
        EncodeOneBin(bac, g->gt1 & 1, &cm->greaterThanOneContextModel[gt1_base + 1]);
        if (g->gt1 & 1)
            EncodeOneBin(bac, level > 2, &cm->greaterThanTwoContextModel[gt2_base]);
        EncodeBypassOneBin(bac, g->sign & 1);
        if (level > 2)
            RemainingCoeffExponentialGolombCode(cabac, level - 3, 0);
        return;
    }
    EncodeLastSignificantXY(cabac, tu->last_x, tu->last_y, size, lg, is_chroma);
    const EB_U32 sig_base = is_chroma ? NUMBER_OF_SIG_FLAG_LUMA_CONTEXT_MODELS : 0;
    const EB_S32 last = tu->last_scan_set;
    EB_U32 c1 = 1, csbf = 0; /* csbf: coded-sub-block flags of the current (bits 0..7) and the previous (bits 16..23) diagonal, by row */
    EB_S32 diag_before = -1;
    for (EB_S32 s = last; s >= 0; s--) {
        const SvtAmdCoeffScanGroup *g = &groups[tu->first_group + (last - s)];
        const uint16_t *lv = levels + g->first_level;
        EB_U32 pattern = csbf & 3;
        if (s) {
            const EB_U32 gy = sub_block_scan[lg - 2][s] >> 4, gx = sub_block_scan[lg - 2][s] & 15;
            if ((EB_S32)(gy + gx) != diag_before)
                csbf <<= 16, diag_before = (EB_S32)(gy + gx);
            if (s != last) {
                pattern = (csbf >> (16 + gy)) & 3;
                EncodeOneBin(bac, g->sigmap != 0, &cm->coeffGroupSigFlagContextModel[(pattern != 0) + is_chroma * NUMBER_OF_COEFF_GROUP_SIG_FLAG_CONTEXT_MODELS]);
                if (!g->sigmap)
                    continue;
            }
            csbf += 1u << gy;
        }
        /* sig_coeff_flag of the positions below the last one (last sub-block) / of all 16 (the others; position 0 is inferred when it is the
         * only one of a sub-block whose coded_sub_block_flag was sent) */
        EB_S32 hi = 15, lo = 0;
        if (s == last)
            hi = (EB_S32)tu->pos_last - 1;
        else if (g->sigmap == 1 && s)
            lo = 1;
        if (hi >= lo) {
            const EB_U8 *map;
            EB_U32 t_off = 0;
            if (lg == 2)
                map = contextIndexMap4[tu->scan_index];
            else {
                t_off = (lg == 3) ? ((tu->scan_index == SCAN_DIAG2 || is_chroma) ? 9 : 15) : (!is_chroma ? 21 : 12);
                t_off += (!is_chroma && s) ? 3 : 0;
                map = contextIndexMap8[tu->scan_index != SCAN_DIAG2][pattern];
            }
            for (EB_S32 k = hi; k >= lo; k--)
                EncodeOneBin(bac, (g->sigmap >> k) & 1, &cm->significanceFlagContextModel[sig_base + ((s | k) ? map[k] + t_off : 0)]);
        }
        /* levels: greater-1 flags of the first 8, one greater-2 flag, signs, remainders */
        EB_U32 n = 0;
        for (EB_U32 m = g->sigmap; m; m &= m - 1)
            n++;
        const EB_U32 n8 = n < GREATER_THAN1_MAX_NUMBER ? n : GREATER_THAN1_MAX_NUMBER;
        const EB_U32 set = ((s && !is_chroma) ? 2 : 0) + (c1 == 0);
        EB_U32 i = 0, first_gt1 = n8, rice = 0;
        c1 = 1;
        for (; i < n8; i++) {
            const EB_U32 f = (g->gt1 >> i) & 1;
            EncodeOneBin(bac, f, &cm->greaterThanOneContextModel[gt1_base + 4 * set + c1]);
            if (f) {
                first_gt1 = i++, c1 = 0;
                break;
            }
            if (c1 < 3)
                c1++;
        }
        for (; i < n8; i++)
            EncodeOneBin(bac, (g->gt1 >> i) & 1, &cm->greaterThanOneContextModel[gt1_base + 4 * set]);
        if (first_gt1 < n8)
            EncodeOneBin(bac, lv[first_gt1] > 2, &cm->greaterThanTwoContextModel[gt2_base + set]);
        EncodeBypassBins(bac, g->sign, n);
        if (first_gt1 < n8) {
            if (lv[first_gt1] >= 3) {
                RemainingCoeffExponentialGolombCode(cabac, lv[first_gt1] - 3u, 0);
                rice = lv[first_gt1] > 3;
            }
            for (i = first_gt1 + 1; i < n8; i++)
                if (lv[i] >= 2)
                    svt_coeff_scan_remainder(cabac, lv[i], 2, &rice);
        }
        for (i = n8; i < n; i++)
            svt_coeff_scan_remainder(cabac, lv[i], 1, &rice);
    }
}
#endif
