/*
 * md_logic.h - the SCALAR decisions of the mode decision of one LCU (Codec/EbProductCodingLoop.c:4691 ModeDecisionLcu and what it calls),
 * restated once in the common subset of C99 and C++ so that the same text is
 *   - the per-coding-unit control code of the HIP kernel (md_kernels.hip; executed by one lane between the data-parallel stages), and
 *   - compiled by gcc into the CPU checker (oracle/svt_oracle_md.c), which runs it against the fixtures recorded from the reference
 *     (tests/golden/md_*.npz) without a GPU.
 * Nothing here touches samples: prediction, SAD, transform, quantiser, rate estimation and reconstruction are the callers' business
 * (device: intra_device.h / txfm_device.h / rate_device.h; checker: the pinned oracle leaves).  Every function names the reference
 * lines it restates (paths relative to /root/reference/Source/Lib).  No reference source is copied: the functions below work on the
 * contract structs of include/svt_hevc_amd.h, not on the reference's contexts.
 */
#ifndef SVT_AMD_MD_LOGIC_H
#define SVT_AMD_MD_LOGIC_H
#include <stdint.h>
#include "../../include/svt_hevc_amd.h"

#ifndef MD_FN
#define MD_FN static inline
#endif

#define MD_INTER 1
#define MD_INTRA 2
#define MD_INVALID_MODE 0xFFu
#define MD_MAX_COST (0xFFFFFFFFFFFFFFFFull >> 1) /* MAX_CU_COST, Codec/EbCodingUnit.h:56 */
#define MD_PLANAR 0
#define MD_DC 1
#define MD_HOR 10
#define MD_VER 26
#define MD_MAX_CAND 48   /* fast-loop candidates of one coding unit: 35 intra modes + 3 MPM + inter candidates */
#define MD_MAX_BUF 8     /* candidate buffers per depth (ProductResetModeDecision, Codec/EbModeDecisionProcess.c:375-392) */

/* ---- GetCodedUnitStats (Codec/EbUtility.c): the 85 leaves of the depth-first scan ---- */
typedef struct MdStats {
    uint8_t depth, size, lg, x, y, num_in_depth, parent, ordinal; /* ordinal: 1..4 among the parent's children (MDSCAN_TO_QUADTREE_ID) */
} MdStats;
MD_FN MdStats md_stats(int leaf)
{
    MdStats s;
    s.depth = 0, s.size = 64, s.lg = 6, s.x = 0, s.y = 0, s.num_in_depth = 0, s.parent = 0, s.ordinal = 1;
    if (leaf == 0)
        return s;
    const int r = leaf - 1, q = r / 21, r32 = r % 21;
    int x = (q & 1) * 32, y = (q >> 1) * 32;
    if (r32 == 0) {
        s.depth = 1, s.size = 32, s.lg = 5, s.parent = 0, s.ordinal = (uint8_t)(q + 1);
    } else {
        const int r2 = r32 - 1, sidx = r2 / 5, r16 = r2 % 5;
        x += (sidx & 1) * 16, y += (sidx >> 1) * 16;
        if (r16 == 0) {
            s.depth = 2, s.size = 16, s.lg = 4, s.parent = (uint8_t)(1 + 21 * q), s.ordinal = (uint8_t)(sidx + 1);
        } else {
            const int e = r16 - 1;
            x += (e & 1) * 8, y += (e >> 1) * 8;
            s.depth = 3, s.size = 8, s.lg = 3, s.parent = (uint8_t)(1 + 21 * q + 1 + 5 * sidx), s.ordinal = (uint8_t)(e + 1);
        }
    }
    s.x = (uint8_t)x, s.y = (uint8_t)y;
    s.num_in_depth = (uint8_t)((y >> s.lg) * (64 >> s.lg) + (x >> s.lg));
    return s;
}
MD_FN int md_depth_offset(int depth) { return depth == 0 ? 85 : depth == 1 ? 21 : depth == 2 ? 5 : 1; } /* DepthOffset / CuOffset */
/* raster-scan index of the OIS / ME tables: cuNumInDepth + me2Nx2NOffset[depth] (Codec/EbDefinitions.h:372) */
MD_FN int md_raster_index(const MdStats *s) { return s->num_in_depth + (s->depth == 0 ? 0 : s->depth == 1 ? 1 : s->depth == 2 ? 5 : 21); }

/* isBottomLeftAvailable / isUpperRightAvailable (Codec/EbAvailability.c:9-69, depth < 4): Z-order availability of the unit's bottom-left
 * and top-right neighbours inside the LCU */
MD_FN int md_bottom_left_ok(const MdStats *s)
{
    const int n = 1 << s->depth, h = s->x >> s->lg, v = s->y >> s->lg;
    int a = h == 0;
    a |= ((h & 1) == 0) && ((v & 1) == 0);
    a |= ((h & 3) == 0) && ((v & 3) == 1);
    return a && v != n - 1;
}
MD_FN int md_top_right_ok(const MdStats *s)
{
    const int n = 1 << s->depth, h = s->x >> s->lg, v = s->y >> s->lg;
    int a = h == n - 1;
    a |= ((h & 1) == 1) && ((v & 1) == 1);
    a |= ((h & 3) == 3) && ((v & 3) == 2);
    return !(a && v != 0);
}

/* ---- per-LCU state: mdLocalCuUnit[] (Codec/EbModeDecisionProcess.h:76-91) + the CodingUnit_t fields the decision writes ---- */
typedef struct MdLocal {
    uint8_t tested, mdc_index;
    uint8_t top_depth, left_depth, top_mode, left_mode; /* 2-bit fields in the reference: 0xFF is stored as 3 */
    uint8_t pad[2];
    uint32_t full_distortion, pad2;                     /* mdLocalCuUnit[].fullDistortion (StopSplitCondition) */
    uint64_t cost;
} MdLocal;
typedef struct MdMv { int16_t x, y; } MdMv;
typedef struct MdCu {
    uint8_t split, pred_mode, intra_luma_mode, ycbf, skip_flag;
    uint8_t left_intra_mode, top_intra_mode; /* PredictionUnit_t.intraLumaLeftMode / TopMode */
    uint8_t skip_ctx;
    uint8_t inter_dir, merge_flag, merge_index, pad; /* PredictionUnit_t.interPredDirectionIndex (3 = intra), .mergeFlag, .mergeIndex */
    MdMv mv[2];                                       /* PredictionUnit_t.mv */
    uint64_t merge_cost, skip_cost;                   /* mdEpPipeLcu[].mergeCost / .skipCost */
    uint64_t y_coeff_bits, y_dist[2], fast_luma_rate; /* mdEpPipeLcu[].yCoeffBits / .yFullDistortion / .fastLumaRate of a merge winner */
    uint32_t ycbf_mask, pad2;                         /* mdEpPipeLcu[].yCbf (candidatePtr->yCbf) */
} MdCu;
typedef struct MdLcuState {
    MdLocal local[SVT_AMD_MD_LEAVES];
    MdCu cu[SVT_AMD_MD_LEAVES];
    uint8_t g8, g16; /* groupOf8x8BlocksCount, groupOf16x16BlocksCount */
} MdLcuState;

/* the neighbour-array entries a coding unit's context generation reads (0xFF: never written / outside the tile) */
typedef struct MdNeighbors {
    uint8_t left_mode, top_mode, left_depth, top_depth, left_skip, top_skip, left_intra, top_intra;
} MdNeighbors;

/* one fast-loop candidate (ModeDecisionCandidate_t, Codec/EbModeDecision.h:92; the fields this revision uses) */
typedef struct MdCand {
    uint8_t type, intra_mode, mpm, dist_ready;
    uint32_t me_dist;
    uint8_t dir, merge_flag, merge_index, mvp_idx[2], pad[3]; /* predictionDirection[0], mergeFlag, mergeIndex, motionVectorPredIdx */
    MdMv mv[2], mvp[2];                                       /* motionVector_{x,y}_L0 / L1, motionVectorPred_{x,y} */
} MdCand;

/* ConstructMdCuArray (Codec/EbProductCodingLoop.c:1290) */
MD_FN void md_construct_cu_array(MdLcuState *S, const SvtAmdMdLcu *L)
{
    int maxCu = 0;
    for (int i = 0; i < L->leaf_count; i++)
        maxCu = L->leaf_index[i] > maxCu ? L->leaf_index[i] : maxCu;
    /* the reference's loop stops one short of the highest leaf (do { } while (cuIdx < maxCuIndex)); that leaf's flags are set by its own
     * test before anything reads them, so resetting it too changes no decision and keeps `tested` = "tested in this LCU" */
    (void)maxCu;
    for (int i = 0; i < SVT_AMD_MD_LEAVES; i++)
        S->local[i].tested = 0, S->cu[i].split = 1;
    S->g8 = S->g16 = 0;
}

/* CodingLoopContextGeneration (Codec/EbRateDistortionCost.c:70-130) */
MD_FN void md_context_generation(MdLcuState *S, int leaf, int cu_y_in_lcu, const MdNeighbors *N)
{
    MdCu *c = &S->cu[leaf];
    c->left_intra_mode = (uint8_t)(N->left_mode != MD_INTRA ? MD_DC : N->left_intra);
    c->top_intra_mode = (uint8_t)(N->top_mode != MD_INTRA ? MD_DC : (cu_y_in_lcu == 0 ? MD_DC : N->top_intra));
    c->skip_ctx = (uint8_t)((N->left_mode == MD_INVALID_MODE ? 0 : N->left_skip == 1) + (N->top_mode == MD_INVALID_MODE ? 0 : N->top_skip == 1));
    MdLocal *l = &S->local[leaf];
    l->left_mode = N->left_mode & 3, l->left_depth = N->left_depth & 3, l->top_mode = N->top_mode & 3, l->top_depth = N->top_depth & 3;
}

/* SplitFlagRate (Codec/EbRateDistortionCost.c:2537-2570); tbMaxDepth = 4 */
MD_FN uint64_t md_split_flag_rate(const SvtAmdMdPicture *P, const MdLcuState *S, int leaf, int splitFlag)
{
    const MdStats st = md_stats(leaf);
    const MdLocal *l = &S->local[leaf];
    const int ctx = (l->left_mode > MD_INTRA ? 0 : l->left_depth > st.depth) + (l->top_mode > MD_INTRA ? 0 : l->top_depth > st.depth);
    const uint64_t rate = st.depth < 3 ? P->rates.splitFlagBits[splitFlag * 3 + ctx] : 0;
    return ((uint64_t)P->full_lambda * rate + (1u << 22)) >> 23;
}

/* DeriveMpmModes (Codec/EbProductCodingLoop.c:872) */
MD_FN void md_mpm_modes(int left, int top, uint32_t mpm[3])
{
    if (left == top) {
        if (left > 1)
            mpm[0] = (uint32_t)left, mpm[1] = (uint32_t)(((left + 29) & 0x1F) + 2), mpm[2] = (uint32_t)(((left - 1) & 0x1F) + 2);
        else
            mpm[0] = MD_PLANAR, mpm[1] = MD_DC, mpm[2] = MD_VER;
    } else {
        mpm[0] = (uint32_t)left, mpm[1] = (uint32_t)top;
        mpm[2] = (left && top) ? MD_PLANAR : ((left + top) < 2 ? MD_VER : MD_DC);
    }
}

/* SetNfl, MDC_STAGE (Codec/EbProductCodingLoop.c:907-961): fullReconSearchCount.  Level 3 looks at picture-analysis detectors this
 * revision does not carry: the caller must not offer such pictures (svt_amd_md_picture_supported). */
MD_FN int md_nfl(const SvtAmdMdPicture *P, const SvtAmdMdLcu *L, int cuSize)
{
    if (P->depth_mode == 0 /* PICT_LCU_SWITCH */ && L->lcu_md_mode == 10 /* LCU_PRED_OPEN_LOOP_1_NFL_DEPTH_MODE */)
        return 1;
    switch (P->nfl_level_md) {
    case 0: return 4;
    case 1: return cuSize == 32 ? 3 : 2;
    case 2: return 2;
    case 4: return cuSize >= 16 ? 2 : 1;
    case 5: return cuSize >= 32 ? 2 : 1;
    default: return 1;
    }
}

MD_FN int md_anti_contouring_valid(int mode) { return mode < 2 || ((mode - 2) & 3) == 0; } /* AntiContouringIntraModeValidityPerDepth */

/* ProductIntraCandidateInjection (Codec/EbModeDecision.c:1105-1536), I pictures and the LCU_COMPLEXITY_STATUS_2 branch; P / B pictures
 * are not offered to this revision.  ois: the LCU's open-loop intra search record.  Returns the candidate count. */
MD_FN int md_intra_candidates(const SvtAmdMdPicture *P, const SvtAmdMdLcu *L, const SvtAmdOisLcuResult *ois, int leaf, const MdStats *st, MdCand *cand)
{
    const uint8_t antiContouringMode[4] = {MD_PLANAR, MD_DC, MD_HOR, MD_VER}; /* first 4 of AntiContouringIntraMode */
    int n = 0;
    const int cuSize = st->size;
    if (P->intra_injection_method == 2) {
        for (int m = 0; m < 35; m++, n++)
            cand[n].type = MD_INTRA, cand[n].intra_mode = (uint8_t)m, cand[n].dist_ready = 0, cand[n].me_dist = 0;
        return n;
    }
    const int isLeftCu = st->x == 0, isTopCu = st->y == 0, limitIntra = P->limit_intra;
    const int skipOis8x8 = P->skip_ois_8x8 && cuSize == 8;
    const int ri = md_raster_index(st);
    if (L->complexity_status_2) {
        if (P->cu8x8_mode == 1) {
            if (limitIntra == 0 || (isLeftCu == 0 && isTopCu == 0)) {
                /* (the reference increments the count once for two writes: the planar candidate overwrites nothing that is counted) */
                cand[n].type = MD_INTRA, cand[n].intra_mode = MD_DC, cand[n].dist_ready = 0, cand[n].me_dist = 0;
                n++;
                cand[n].type = MD_INTRA, cand[n].intra_mode = MD_PLANAR, cand[n].dist_ready = 0, cand[n].me_dist = 0;
            }
        } else if (skipOis8x8) {
            if (limitIntra == 0 || (isLeftCu == 0 && isTopCu == 0)) {
                cand[n].type = MD_INTRA, cand[n].intra_mode = MD_PLANAR, cand[n].dist_ready = 0, cand[n].me_dist = 0;
                n++;
            }
        } else {
            const int total = ois->total_intra_luma_mode[ri];
            for (int k = 0; k < total; k++) {
                const uint32_t w = ois->candidate[ri][k];
                const int mode = (int)(w >> 24);
                if ((limitIntra == 0 || (isLeftCu == 0 && isTopCu == 0)) && (mode == MD_PLANAR || mode == MD_DC)) {
                    cand[n].type = MD_INTRA, cand[n].intra_mode = (uint8_t)mode, cand[n].dist_ready = (uint8_t)((w >> 20) & 1), cand[n].me_dist = w & 0xFFFFFu;
                    n++;
                }
            }
        }
        return n;
    }
    if (st->depth == 0)
        return 0;
    if (P->slice_type != 2) { /* P / B pictures (:1311-1526); limitOisToDcModeFlag (encMode >= 10) is not offered to this revision */
        if (cuSize == 32 || (cuSize >= 16 && P->cu16x16_mode == 0 && P->enc_mode < 11)) {
            const int total = ois->total_intra_luma_mode[ri];
            for (int k = 0; k < total; k++) {
                const uint32_t w = ois->candidate[ri][k];
                const int mode = (int)(w >> 24);
                if (md_anti_contouring_valid(mode)) {
                    cand[n].type = MD_INTRA, cand[n].intra_mode = (uint8_t)mode, cand[n].dist_ready = (uint8_t)((w >> 20) & 1), cand[n].me_dist = w & 0xFFFFFu;
                    n++;
                    if (limitIntra && ((isLeftCu && mode != (cuSize < 32 ? 27 : MD_VER)) || (isTopCu && mode != (cuSize < 32 ? 9 : MD_HOR))))
                        n--;
                }
            }
        } else if (cuSize == 16) {
            if (limitIntra == 0 || (isLeftCu == 0 && isTopCu == 0)) {
                cand[n].type = MD_INTRA, cand[n].intra_mode = MD_DC, cand[n].dist_ready = 0, cand[n].me_dist = 0, n++;
                cand[n].type = MD_INTRA, cand[n].intra_mode = MD_PLANAR, cand[n].dist_ready = 0, cand[n].me_dist = 0, n++;
            }
        } else if (skipOis8x8) { /* both 8x8 modes */
            if (limitIntra == 0 || (isLeftCu == 0 && isTopCu == 0))
                cand[n].type = MD_INTRA, cand[n].intra_mode = MD_PLANAR, cand[n].dist_ready = 0, cand[n].me_dist = 0, n++;
        } else if (P->cu8x8_mode == 1) {
            if (L->is_complete) { /* the parent 16x16 unit's open-loop candidates */
                const MdStats ps = md_stats(st->parent);
                const int pri = md_raster_index(&ps), total = ois->total_intra_luma_mode[pri];
                for (int k = 0; k < total; k++) {
                    const uint32_t w = ois->candidate[pri][k];
                    const int mode = (int)(w >> 24);
                    if (md_anti_contouring_valid(mode)) {
                        cand[n].type = MD_INTRA, cand[n].intra_mode = (uint8_t)mode, cand[n].dist_ready = 0, cand[n].me_dist = w & 0xFFFFFu;
                        n++;
                        if (limitIntra && ((isLeftCu && mode != 27) || (isTopCu && mode != 9)))
                            n--;
                    }
                }
            } else if (limitIntra == 0 || (isLeftCu == 0 && isTopCu == 0)) {
                cand[n].type = MD_INTRA, cand[n].intra_mode = MD_DC, cand[n].dist_ready = 0, cand[n].me_dist = 0, n++;
            }
        } else {
            const int total = ois->total_intra_luma_mode[ri];
            for (int k = 0; k < total; k++) {
                const uint32_t w = ois->candidate[ri][k];
                const int mode = (int)(w >> 24);
                if (!P->intra8x8_restriction_inter_slice || md_anti_contouring_valid(mode)) {
                    cand[n].type = MD_INTRA, cand[n].intra_mode = (uint8_t)mode, cand[n].dist_ready = (uint8_t)((w >> 20) & 1), cand[n].me_dist = w & 0xFFFFFu;
                    n++;
                    if (limitIntra && ((isLeftCu && mode != 27) || (isTopCu && mode != 9)))
                        n--;
                }
            }
        }
        return n;
    }
    const int contouring = L->contouring_class[(leaf - 1) / 21];
    if (cuSize == 32) {
        if (P->intra_injection_method == 1 && contouring == 0) {
            for (int m = 0; m < 35; m++, n++)
                cand[n].type = MD_INTRA, cand[n].intra_mode = (uint8_t)m, cand[n].dist_ready = 0, cand[n].me_dist = 0;
        } else {
            for (int m = 0; m < 4; m++, n++)
                cand[n].type = MD_INTRA, cand[n].intra_mode = antiContouringMode[m], cand[n].dist_ready = 0, cand[n].me_dist = 0;
        }
    } else if (cuSize == 16) {
        if (P->intra_injection_method == 1 && contouring == 0) {
            for (int m = 0; m < 35; m++, n++)
                cand[n].type = MD_INTRA, cand[n].intra_mode = (uint8_t)m, cand[n].dist_ready = 0, cand[n].me_dist = 0;
        } else {
            const int total = ois->total_intra_luma_mode[ri];
            for (int k = 0; k < total; k++) {
                const uint32_t w = ois->candidate[ri][k];
                if (md_anti_contouring_valid((int)(w >> 24))) {
                    cand[n].type = MD_INTRA, cand[n].intra_mode = (uint8_t)(w >> 24), cand[n].dist_ready = 0, cand[n].me_dist = w & 0xFFFFFu;
                    n++;
                }
            }
        }
    } else {
        if (P->intra_injection_method == 1) {
            for (int m = 0; m < 35; m++, n++)
                cand[n].type = MD_INTRA, cand[n].intra_mode = (uint8_t)m, cand[n].dist_ready = 0, cand[n].me_dist = 0;
        } else {
            const int total = ois->total_intra_luma_mode[ri];
            for (int k = 0; k < total; k++, n++) {
                const uint32_t w = ois->candidate[ri][k];
                cand[n].type = MD_INTRA, cand[n].intra_mode = (uint8_t)(w >> 24), cand[n].dist_ready = 0, cand[n].me_dist = w & 0xFFFFFu;
            }
        }
    }
    return n;
}

/* ProductMpmCandidatesInjection (Codec/EbModeDecision.c:1703-1788; LIMITINRA_MPM_PATCH is not defined).  Returns the new candidate
 * count; *bufferTotal grows by one per most probable mode searched. */
MD_FN int md_mpm_injection(const SvtAmdMdPicture *P, const SvtAmdMdLcu *L, const MdStats *st, MdCand *cand, int n, int *bufferTotal, const uint32_t mpm[3])
{
    const int mpmSearch = P->mpm_search && !L->restrict_intra_global_motion;
    if (mpmSearch && st->depth != 0) {
        const int fast = n;
        for (int i = 0; i < fast; i++)
            cand[i].mpm = 0;
        for (int k = 0; k < P->mpm_search_candidate; k++) {
            int present = 0;
            for (int i = 0; i < fast; i++)
                if (cand[i].type == MD_INTRA && !present && cand[i].intra_mode == mpm[k])
                    cand[i].mpm = 1, ++*bufferTotal, present = 1;
            if (!present) {
                cand[n].type = MD_INTRA, cand[n].intra_mode = (uint8_t)mpm[k], cand[n].dist_ready = 0, cand[n].me_dist = 0, cand[n].mpm = 1;
                n++, ++*bufferTotal;
            }
        }
    } else {
        for (int i = 0; i < n; i++)
            cand[i].mpm = 0;
    }
    return n;
}

/* Intra2Nx2NFastCostIsliceOpt (Codec/EbRateDistortionCost.c:440-491).  chromaWeight = ChromaWeightFactorLd[qp] (chroma distortion is
 * 0 without chroma in the fast loop).  *fastLumaRate: candidatePtr->fastLumaRate. */
MD_FN uint64_t md_intra_fast_cost_islice(const SvtAmdMdPicture *P, const MdStats *st, const MdCu *c, int lumaMode, uint64_t lumaDistortion,
                                         uint64_t *fastLumaRate)
{
    const uint64_t chromaRate = 9732;
    uint64_t lumaRate = st->depth == 3 ? 24752 : 0;
    lumaRate += (lumaMode == c->left_intra_mode || lumaMode == c->top_intra_mode) ? 57520 : 206378;
    *fastLumaRate = lumaRate;
    const uint64_t lumaSad = lumaDistortion << 8;
    const uint64_t rate = ((uint64_t)P->fast_lambda * (lumaRate + chromaRate) + (1u << 22)) >> 23;
    return lumaSad + rate;
}

/* the candidate buffers of one depth: fastCostArray / fullCostArray + which candidate sits in each (ProductPerformFastLoop's second
 * loop, Codec/EbProductCodingLoop.c:1990-2179).  costs[i] / evaluated[i]: fast cost of candidate i and whether the loop evaluates it
 * (not distortion-ready, or the best of the first loop, or singleFastLoopFlag). */
typedef struct MdBuffers {
    uint64_t fast_cost[MD_MAX_BUF], full_cost[MD_MAX_BUF];
    int16_t cand[MD_MAX_BUF]; /* candidatePtr of the buffer: index into the candidate array, -1 = never assigned */
    int16_t pred[MD_MAX_BUF]; /* the candidate whose luma prediction the buffer holds (the last one the loop evaluated there; a candidate
                               * the loop does not evaluate takes the buffer's candidatePtr, not its samples), -1 = none of this unit */
    int evaluated_count;      /* secondFastCostSearchCandidateTotalCount */
} MdBuffers;
/* evaluated[i]: 1 = evaluated, 3 = evaluated without a luma prediction (the open-loop intra candidate that won the first loop) */
MD_FN void md_fast_loop_buffers(MdBuffers *B, int width, int maxBuffers, int ncand, const uint64_t *costs, const uint8_t *evaluated)
{
    for (int i = 0; i < MD_MAX_BUF; i++)
        B->fast_cost[i] = B->full_cost[i] = ~0ull, B->cand[i] = -1, B->pred[i] = -1; /* EbHevcProductCodingLoopInitFastLoop :1586-1596 (i < width) */
    (void)width;
    B->evaluated_count = 0;
    int highest = 0;
    for (int idx = ncand - 1; idx >= 0; idx--) {
        B->cand[highest] = (int16_t)idx;
        if (evaluated[idx]) {
            B->fast_cost[highest] = costs[idx];
            B->evaluated_count++;
            if (!(evaluated[idx] & 2))
                B->pred[highest] = (int16_t)idx;
        }
        if (idx) { /* the buffer with the highest cost (an unused one first) takes the next candidate */
            highest = 0;
            int b = 1;
            do {
                const uint64_t hc = B->fast_cost[highest];
                if (hc == ~0ull)
                    break;
                if (B->fast_cost[b] > hc)
                    highest = b;
            } while (++b < maxBuffers);
        }
    }
}

/* PreModeDecision (Codec/EbModeDecision.c:300-383).  types[b]: candidate type of buffer b.  Returns fullCandidateTotalCount. */
MD_FN int md_pre_mode_decision(const MdBuffers *B, const uint8_t *types, int bufferTotalCount, int sameFastFull, uint8_t *best)
{
    int fullRecon = sameFastFull ? (bufferTotalCount < 1 ? 1 : bufferTotalCount) : (bufferTotalCount - 1 < 1 ? 1 : bufferTotalCount - 1);
    uint64_t highestCost = B->fast_cost[0];
    int highestIdx = 0, k = 0;
    if (bufferTotalCount > 1) {
        if (sameFastFull) {
            for (int i = 0; i < bufferTotalCount; i++)
                best[k++] = (uint8_t)i;
        } else {
            for (int i = 1; i < bufferTotalCount; i++)
                if (B->fast_cost[i] >= highestCost)
                    highestCost = B->fast_cost[i], highestIdx = i;
            for (int i = 0; i < bufferTotalCount; i++)
                if (i != highestIdx)
                    best[k++] = (uint8_t)i;
        }
    } else {
        best[0] = 0;
    }
    for (int i = 0; i < fullRecon - 1; i++) /* inter candidates first */
        for (int j = i + 1; j < fullRecon; j++)
            if (types[best[i]] == MD_INTRA && types[best[j]] == MD_INTER) {
                const uint8_t t = best[i];
                best[i] = best[j], best[j] = t;
            }
    return fullRecon;
}

/* IntraFullLumaCostIslice (Codec/EbRateDistortionCost.c:963-1040): transformSize == cuSize for the units this revision covers */
MD_FN uint64_t md_intra_full_luma_cost_islice(const SvtAmdMdPicture *P, int transformLg, uint32_t ycbf, uint64_t fastLumaRate, uint64_t yDistortion0,
                                              uint64_t yCoeffBits)
{
    const uint32_t transSubDivFlagCtx = (uint32_t)(5 - transformLg);
    uint64_t lumaRate = P->rates.transSubDivFlagBits[transSubDivFlagCtx] + P->rates.lumaCbfBits[(ycbf & 1) * 5 + 1];
    lumaRate += fastLumaRate;
    const uint64_t coeffRate = yCoeffBits << 15, distortion = yDistortion0 << 8, lambda = P->full_lambda;
    return distortion + (((lambda * coeffRate + lambda * lumaRate) + (1u << 22)) >> 23);
}

/* CheckHighCostPartition (Codec/EbProductCodingLoop.c:1164-1231): after a child that is a leaf of the tree, the children tested so far
 * already cost more than their (tested) parent -> the parent wins and the remaining children are not tested.  Returns the parent's
 * leaf index, or -1.  enableExitPartitioning is off below encMode 10 (EbEncDecProcess.c:2217-2228). */
MD_FN int md_check_high_cost_partition(const SvtAmdMdPicture *P, const SvtAmdMdLcu *L, const MdLcuState *S, int leaf)
{
    const MdStats st = md_stats(leaf);
    if (!L->is_complete || st.depth == 0)
        return -1;
    if (st.ordinal >= 4 || S->cu[leaf].split)
        return -1;
    const int parent = st.parent;
    if (!S->local[parent].tested)
        return -1;
    const uint64_t parentCost = S->local[parent].cost + md_split_flag_rate(P, S, parent, 0);
    uint64_t children = 0;
    int it = leaf;
    for (int k = 0; k < st.ordinal; k++) {
        children += S->local[it].cost;
        it -= md_depth_offset(st.depth);
    }
    return children > parentCost ? parent : -1;
}

/* ProductPerformInterDepthDecision (Codec/EbFullLoop.c:1461-1776; stop_split: StopSplitCondition, md_stop_split below)
 * and, with exit_parent != 0, ExitInterDepthDecision (:1070-1378).  Returns lastCuIndex. */
MD_FN int md_inter_depth_decision(const SvtAmdMdPicture *P, MdLcuState *S, int leaf, int lcu_x, int lcu_y, int exit_parent, int stop_split)
{
    int last = leaf;
    const MdStats cur = md_stats(leaf);
    int cuX = lcu_x + cur.x, cuY = lcu_y + cur.y;
    int d1 = leaf, d2 = leaf;
    if (exit_parent) {
        S->local[leaf].cost += md_split_flag_rate(P, S, leaf, 0);
        if (cur.depth == 0)
            S->g16 = 0;
        else if (cur.depth == 1)
            S->g16++, S->g8 = 0;
        else if (cur.depth == 2)
            S->g8++;
    } else if (S->cu[leaf].split == 0 || stop_split) { /* lastDepthFlag || stopSplitFlag */
        S->cu[leaf].split = 0;
        if (cur.depth == 1)
            S->g16++;
        else if (cur.depth == 2)
            S->g8++;
    }
    /* stage 0: depth 2 vs depth 3 */
    if (((cuX >> 3) & 1) && ((cuY >> 3) & 1)) {
        S->g8++;
        const int left = leaf - 1, top = left - 1, topLeft = top - 1;
        d2 = topLeft - 1;
        S->local[d2].left_mode = S->local[topLeft].left_mode, S->local[d2].left_depth = S->local[topLeft].left_depth;
        S->local[d2].top_mode = S->local[topLeft].top_mode, S->local[d2].top_depth = S->local[topLeft].top_depth;
        const uint64_t rateN = md_split_flag_rate(P, S, d2, 0);
        if (!S->local[d2].tested)
            S->local[d2].cost = MD_MAX_COST;
        const uint64_t costN = S->local[d2].cost + rateN;
        const uint64_t costN1 = S->local[leaf].cost + S->local[left].cost + S->local[top].cost + S->local[topLeft].cost + md_split_flag_rate(P, S, d2, 1);
        if (costN <= costN1)
            S->cu[d2].split = 0, S->local[d2].cost = costN, last = d2;
        else
            S->local[d2].cost = costN1;
    }
    /* stage 1: depth 1 vs depth 2 */
    const MdStats s2 = md_stats(d2);
    cuX = lcu_x + s2.x, cuY = lcu_y + s2.y;
    if ((((cuX >> 3) & 2) == 2) && (((cuY >> 3) & 2) == 2) && S->g8 == 4) {
        S->g8 = 0, S->g16++;
        const int left = d2 - 5, top = left - 5, topLeft = top - 5;
        d1 = topLeft - 1;
        const int tmp = d2 - 16; /* the index the reference copies the contexts to before it recomputes the candidate (:1596, :1605-1608) */
        S->local[tmp].left_mode = S->local[topLeft].left_mode, S->local[tmp].left_depth = S->local[topLeft].left_depth;
        S->local[tmp].top_mode = S->local[topLeft].top_mode, S->local[tmp].top_depth = S->local[topLeft].top_depth;
        if (md_stats(d1).depth == 1) {
            const uint64_t rateN = md_split_flag_rate(P, S, d1, 0);
            if (!S->local[d1].tested)
                S->local[d1].cost = MD_MAX_COST;
            const uint64_t costN = S->local[d1].cost + rateN;
            const uint64_t costN1 = S->local[d2].cost + S->local[left].cost + S->local[top].cost + S->local[topLeft].cost + md_split_flag_rate(P, S, d1, 1);
            if (costN <= costN1)
                S->cu[d1].split = 0, S->local[d1].cost = costN, last = d1;
            else
                S->local[d1].cost = costN1;
        }
    }
    /* stage 2: depth 0 vs depth 1 (P / B pictures only: no 64x64 candidate in I pictures) */
    if (P->slice_type != 2 && (((cuX >> 3) & 4) == 4) && (((cuY >> 3) & 4) == 4) && S->g16 == 4) {
        S->g16 = 0;
        const int left = d1 - 21, top = left - 21, topLeft = top - 21;
        const int tmp = d1 - 64;
        if (tmp >= 0) {
            S->local[tmp].left_mode = S->local[topLeft].left_mode, S->local[tmp].left_depth = S->local[topLeft].left_depth;
            S->local[tmp].top_mode = S->local[topLeft].top_mode, S->local[tmp].top_depth = S->local[topLeft].top_depth;
        }
        const int d0 = topLeft - 1;
        if (d0 == 0) {
            const uint64_t rateN = md_split_flag_rate(P, S, d0, 0);
            if (!S->local[d0].tested)
                S->local[d0].cost = MD_MAX_COST;
            const uint64_t costN = S->local[d0].cost + rateN;
            const uint64_t costN1 = S->local[d1].cost + S->local[left].cost + S->local[top].cost + S->local[topLeft].cost + md_split_flag_rate(P, S, d0, 1);
            if (costN <= costN1)
                S->cu[d0].split = 0, last = d0;
        }
    }
    return last;
}

/* CalculateNextCuIndex (Codec/EbProductCodingLoop.c:1261-1289) */
MD_FN int md_next_cu_step(const SvtAmdMdLcu *L, int cuIdx, int depth)
{
    const int next = depth == 0 ? L->leaf_index[L->leaf_count - 1] + 1 : L->leaf_index[cuIdx] + md_depth_offset(depth);
    int step = 1;
    for (int i = cuIdx + 1; i < L->leaf_count; i++) {
        if (L->leaf_index[i] < next)
            step++;
        else
            break;
    }
    return step;
}

/* ================================================ P / B pictures ================================================ */
#define MD_L0 0
#define MD_L1 1
#define MD_BI 2
/* MvUnit_t (Codec/EbDefinitions.h) of a neighbouring position + whether the candidate derivation may use it */
typedef struct MdMvUnit {
    MdMv mv[2];
    uint8_t dir, avail, pad[2];
} MdMvUnit;
/* the five spatial neighbours of a unit, in the order the reference names them */
enum { MD_A0 = 0, MD_A1, MD_B0, MD_B1, MD_B2 };
typedef struct MdMergeCand {
    MdMv mv[2];
    uint8_t dir, pad[3];
} MdMergeCand;
typedef struct MdInterLists {
    MdMv amvp[2][2];       /* firstPuAMVPCandArray_{x,y}[list][idx] */
    uint8_t amvp_count[2]; /* firstPuNumAvailableAMVPCand */
    uint8_t merge_count, pad;
    MdMergeCand merge[5];  /* interPredictionPtr->mvMergeCandidateArray */
} MdInterLists;

MD_FN int md_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

/* ScaleMV (Codec/EbAdaptiveMotionVectorPrediction.c:28-52) */
MD_FN void md_scale_mv(uint64_t curPoc, uint64_t targetRefPoc, uint64_t colPoc, uint64_t colRefPoc, MdMv *mv)
{
    int16_t td = (int16_t)(colPoc - colRefPoc), tb = (int16_t)(curPoc - targetRefPoc);
    if (td != tb) {
        tb = (int16_t)md_clip3(-128, 127, tb), td = (int16_t)md_clip3(-128, 127, td);
        const int16_t temp = (int16_t)((0x4000 + ((td >> 1) < 0 ? -(td >> 1) : (td >> 1))) / td);
        const int16_t scale = (int16_t)md_clip3(-4096, 4095, (tb * temp + 32) >> 6);
        mv->x = (int16_t)md_clip3(-32768, 32767, (scale * mv->x + 127 + (scale * mv->x < 0)) >> 8);
        mv->y = (int16_t)md_clip3(-32768, 32767, (scale * mv->y + 127 + (scale * mv->y < 0)) >> 8);
    }
}

/* GetNonScalingSpatialAMVP_V2 (:172-243): the neighbour's vector that points to the target reference picture, if any */
MD_FN int md_amvp_non_scaling(const MdMvUnit *u, int targetList, uint64_t targetPoc, const uint64_t refPoc[2], MdMv *out)
{
    int avail;
    if (u->dir == MD_L0 || u->dir == MD_L1) {
        avail = targetPoc == refPoc[u->dir];
        if (avail)
            *out = u->mv[u->dir];
    } else {
        avail = targetPoc == refPoc[targetList];
        if (avail)
            *out = u->mv[targetList];
        else {
            avail = targetPoc == refPoc[1 - targetList];
            *out = u->mv[1 - targetList]; /* written either way, as the reference does */
        }
    }
    return avail;
}
/* GetScalingSpatialAMVP_V2 (:247-288): always available */
MD_FN void md_amvp_scaling(const MdMvUnit *u, int targetList, uint64_t targetPoc, uint64_t curPoc, const uint64_t refPoc[2], MdMv *out)
{
    const int list = u->dir == MD_BI ? targetList : u->dir;
    *out = u->mv[list];
    md_scale_mv(curPoc, targetPoc, curPoc, refPoc[list], out);
}

/* position of the temporal candidate in the co-located picture's motion field (GenerateL0L1AmvpMergeLists :2416-2450, repeated :2573, :2868) */
typedef struct MdTmvpPos { uint8_t bottom_right, lcu_offset, unit, pad; } MdTmvpPos;
MD_FN MdTmvpPos md_tmvp_position(const SvtAmdMdPicture *P, const SvtAmdTmvpLcu *map, int ox, int oy, int size)
{
    MdTmvpPos t;
    t.bottom_right = 0, t.lcu_offset = 0, t.unit = 0, t.pad = 0;
    if (!(ox + size >= P->width || oy + size >= P->height || ((oy & 63) + size) >= 64)) {
        const int brx = (ox & 63) + size;
        const int off = brx >> 6, bx = (brx & 63) >> 4, by = ((oy & 63) + size) >> 4;
        const int unit = by * 4 + bx;
        if (map[off].available[unit] == 1)
            t.bottom_right = 1, t.lcu_offset = (uint8_t)off, t.unit = (uint8_t)unit;
    }
    if (!t.bottom_right)
        t.unit = (uint8_t)(((((oy & 63) + (size >> 1)) >> 4) * 4) + (((ox & 63) + (size >> 1)) >> 4));
    return t;
}
/* GetTemporalMVP_V2 (:1394-1450) / one list of GetTemporalMVPBPicture_V2 (:1592-1745) / GetTemporalMVP (:1275-1389) */
MD_FN int md_temporal_mvp(const SvtAmdMdInter *X, const SvtAmdTmvpLcu *map, MdTmvpPos t, int targetList, uint64_t targetPoc, MdMv *out)
{
    int colList = X->is_low_delay ? targetList : 1 - X->colocated_pu_ref_list;
    const SvtAmdTmvpLcu *m = &map[t.bottom_right ? t.lcu_offset : 0];
    if (!t.bottom_right && !m->available[t.unit])
        return 0;
    colList = m->pred_dir[t.unit] == MD_BI ? colList : m->pred_dir[t.unit];
    out->x = m->mv[colList][t.unit][0], out->y = m->mv[colList][t.unit][1];
    md_scale_mv(X->picture_number, targetPoc, X->colocated_poc, m->ref_poc[colList][t.unit], out);
    return 1;
}

MD_FN int md_mv_differs(const MdMvUnit *a, const MdMvUnit *b, int bslice)
{
    if (!bslice)
        return a->mv[0].x != b->mv[0].x || a->mv[0].y != b->mv[0].y;
    return a->dir != b->dir || a->mv[0].x != b->mv[0].x || a->mv[0].y != b->mv[0].y || a->mv[1].x != b->mv[1].x || a->mv[1].y != b->mv[1].y;
}

/* SetNmm, MDC_STAGE (Codec/EbProductCodingLoop.c:1110-1123): mvMergeSkipModeCount */
MD_FN int md_nmm(const SvtAmdMdPicture *P, int cuSize) { return P->nmm_level_md == 0 ? 5 : (P->nmm_level_md == 1 ? (cuSize == 32 ? 3 : 2) : 2); }

/* GenerateL0L1AmvpMergeLists (Codec/EbAdaptiveMotionVectorPrediction.c:2117-3005), generateAmvpTableMd on.  nb[]: the spatial
 * neighbours with the availability the reference derives (:2256-2340: scan order, array bound, inter mode, tile edges); map: the
 * co-located picture's motion field at this LCU (entry 1 = the LCU to the right), NULL when the temporal candidate is off. */
/* parts: 1 = the AMVP candidates of list 0, 2 = of list 1, 4 = the merge candidates - three computations that share their inputs and nothing else (the kernel
 * runs them on three waves); 7 = all */
MD_FN void md_amvp_merge_lists_parts(const SvtAmdMdPicture *P, const SvtAmdMdInter *X, const MdMvUnit nb[5], const SvtAmdTmvpLcu *map, int ox, int oy,
                                     int size, int totalMerge, MdInterLists *o, int parts)
{
    const int bslice = P->slice_type == 0;
    const MdMvUnit *A0 = &nb[MD_A0], *A1 = &nb[MD_A1], *B0 = &nb[MD_B0], *B1 = &nb[MD_B1], *B2 = &nb[MD_B2];
    MdTmvpPos tp;
    tp.bottom_right = 0, tp.lcu_offset = 0, tp.unit = 0, tp.pad = 0;
    if (map)
        tp = md_tmvp_position(P, map, ox, oy, size);
    for (int list = 0; list < (bslice ? 2 : 1); list++) {
        if (!(parts & (1 << list)))
            continue;
        const uint64_t targetPoc = X->ref_poc[list];
        MdMv *c = o->amvp[list];
        int num = 0, ax = 0;
        /* GetSpatialMVPPosAx_V3 (:470-572) */
        if (A0->avail)
            ax = md_amvp_non_scaling(A0, list, targetPoc, X->ref_poc, &c[num]);
        if (!ax && A1->avail)
            ax = md_amvp_non_scaling(A1, list, targetPoc, X->ref_poc, &c[num]);
        if (!ax && (A0->avail || A1->avail)) {
            md_amvp_scaling(A0->avail ? A0 : A1, list, targetPoc, X->picture_number, X->ref_poc, &c[num]);
            ax = 1;
        }
        num += ax;
        /* GetNonScalingSpatialMVPPosBx_V3 (:926-1094) */
        int bx = 0;
        if (B0->avail)
            bx = md_amvp_non_scaling(B0, list, targetPoc, X->ref_poc, &c[num]);
        if (!bx && B1->avail)
            bx = md_amvp_non_scaling(B1, list, targetPoc, X->ref_poc, &c[num]);
        if (!bx && B2->avail)
            bx = md_amvp_non_scaling(B2, list, targetPoc, X->ref_poc, &c[num]);
        num += bx;
        /* GetScalingSpatialMVPPosBx_V3 (:1099-1270): the first available of B0, B1, B2 */
        if (!ax && (B0->avail || B1->avail || B2->avail)) {
            md_amvp_scaling(B0->avail ? B0 : (B1->avail ? B1 : B2), list, targetPoc, X->picture_number, X->ref_poc, &c[num]);
            num++;
        }
        if (num == 2 && c[0].x == c[1].x && c[0].y == c[1].y)
            num = 1;
        if (map && num < 2)
            num += md_temporal_mvp(X, map, tp, list, targetPoc, &c[num]);
        if (num < 1 || (num == 1 && c[0].x != 0 && c[0].y != 0)) {
            c[num].x = 0, c[num].y = 0;
            num++;
        }
        o->amvp_count[list] = (uint8_t)num;
    }
    if (!(parts & 4))
        return;
    /* merge candidates (:2649-2990) */
    MdMergeCand *m = o->merge;
    int idx = 0;
    do {
        if (A1->avail)
            m[idx].dir = bslice ? A1->dir : MD_L0, m[idx].mv[0] = A1->mv[0], m[idx].mv[1] = A1->mv[1], idx++;
        if (idx == totalMerge)
            break;
        if (B1->avail && (!A1->avail || md_mv_differs(B1, A1, bslice)))
            m[idx].dir = bslice ? B1->dir : MD_L0, m[idx].mv[0] = B1->mv[0], m[idx].mv[1] = B1->mv[1], idx++;
        if (idx == totalMerge)
            break;
        if (B0->avail && (!B1->avail || md_mv_differs(B0, B1, bslice)))
            m[idx].dir = bslice ? B0->dir : MD_L0, m[idx].mv[0] = B0->mv[0], m[idx].mv[1] = B0->mv[1], idx++;
        if (idx == totalMerge)
            break;
        if (A0->avail && (!A1->avail || md_mv_differs(A0, A1, bslice)))
            m[idx].dir = bslice ? A0->dir : MD_L0, m[idx].mv[0] = A0->mv[0], m[idx].mv[1] = A0->mv[1], idx++;
        if (idx == totalMerge)
            break;
        if (idx < 4 && B2->avail && (!A1->avail || md_mv_differs(B2, A1, bslice)) && (!B1->avail || md_mv_differs(B2, B1, bslice)))
            m[idx].dir = bslice ? B2->dir : MD_L0, m[idx].mv[0] = B2->mv[0], m[idx].mv[1] = B2->mv[1], idx++;
        if (idx == totalMerge)
            break;
        if (map) { /* temporal candidate */
            MdMv t0, t1;
            t1.x = t1.y = 0;
            if (bslice) {
                if (md_temporal_mvp(X, map, tp, 0, X->ref_poc[0], &t0)) {
                    md_temporal_mvp(X, map, tp, 1, X->ref_poc[1], &t1);
                    m[idx].dir = MD_BI, m[idx].mv[0] = t0, m[idx].mv[1] = t1, idx++;
                }
            } else if (md_temporal_mvp(X, map, tp, 0, X->ref_poc[0], &t0) && idx < 5) {
                m[idx].dir = MD_L0, m[idx].mv[0] = t0, m[idx].mv[1] = t1, idx++;
            }
        }
        if (idx == totalMerge)
            break;
        if (bslice) { /* combined bi-predictive candidates: mvMergeCandIndexArrayForFillingUp (:20-23) */
            const uint8_t l0c[12] = {0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3}, l1c[12] = {1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2};
            const int loopEnd = idx * (idx - 1);
            for (int f = 0; idx < totalMerge && f < loopEnd; f++) {
                const MdMergeCand *c0 = &m[l0c[f]], *c1 = &m[l1c[f]];
                if (((c0->dir + 1) & 1) && ((c1->dir + 1) & 2) &&
                    (X->ref_poc[0] != X->ref_poc[1] || c0->mv[0].x != c1->mv[1].x || c0->mv[0].y != c1->mv[1].y)) {
                    const MdMv v0 = c0->mv[0], v1 = c1->mv[1];
                    m[idx].dir = MD_BI, m[idx].mv[0] = v0, m[idx].mv[1] = v1, idx++;
                }
            }
            if (idx == totalMerge)
                break;
        }
        for (int r = 0; r < 5 && idx < totalMerge; r++) { /* zero vectors */
            m[idx].dir = bslice ? MD_BI : MD_L0, m[idx].mv[0].x = m[idx].mv[0].y = m[idx].mv[1].x = m[idx].mv[1].y = 0;
            idx++;
        }
    } while (0);
    o->merge_count = (uint8_t)idx;
}
MD_FN void md_amvp_merge_lists(const SvtAmdMdPicture *P, const SvtAmdMdInter *X, const MdMvUnit nb[5], const SvtAmdTmvpLcu *map, int ox, int oy,
                               int size, int totalMerge, MdInterLists *o)
{
    md_amvp_merge_lists_parts(P, X, nb, map, ox, oy, size, totalMerge, o, 7);
}

/* ClipMV (:54-72): the bounds are computed in 32-bit unsigned arithmetic and narrowed to 16 bits, as written */
MD_FN void md_clip_mv(const SvtAmdMdPicture *P, uint32_t ox, uint32_t oy, MdMv *mv)
{
    const int16_t xlo = (int16_t)((1u - ox - 8u - 64u) << 2), xhi = (int16_t)(((uint32_t)P->width + 8u - ox - 1u) << 2);
    const int16_t ylo = (int16_t)((1u - oy - 8u - 64u) << 2), yhi = (int16_t)(((uint32_t)P->height + 8u - oy - 1u) << 2);
    mv->x = (int16_t)md_clip3(xlo, xhi, mv->x), mv->y = (int16_t)md_clip3(ylo, yhi, mv->y);
}
/* ChooseMVPIdx_V2 (Codec/EbInterPrediction.c:1126-1327) for the lists the candidate uses */
MD_FN void md_choose_mvp(const SvtAmdMdPicture *P, uint32_t ox, uint32_t oy, const MdInterLists *T, MdCand *c)
{
    for (int list = 0; list < 2; list++) {
        if (!(c->dir == MD_BI || c->dir == list))
            continue;
        md_clip_mv(P, ox, oy, &c->mv[list]);
        const MdMv *a = T->amvp[list];
        int idx = 0;
        if (T->amvp_count[list] == 2) {
            const uint32_t d0 = (uint32_t)(a[0].x > c->mv[list].x ? a[0].x - c->mv[list].x : c->mv[list].x - a[0].x) +
                                (uint32_t)(a[0].y > c->mv[list].y ? a[0].y - c->mv[list].y : c->mv[list].y - a[0].y);
            const uint32_t d1 = (uint32_t)(a[1].x > c->mv[list].x ? a[1].x - c->mv[list].x : c->mv[list].x - a[1].x) +
                                (uint32_t)(a[1].y > c->mv[list].y ? a[1].y - c->mv[list].y : c->mv[list].y - a[1].y);
            idx = d0 <= d1 ? 0 : 1;
        } else if (T->amvp_count[list] > 2) {
            continue;
        }
        c->mvp_idx[list] = (uint8_t)idx, c->mvp[list] = a[idx];
    }
}

/* Me2Nx2NCandidatesInjection (Codec/EbModeDecision.c:446-566) with sub-sample motion and unrestricted motion vectors (neither RoundMv
 * nor LimitMvOverBound), then ProductMergeSkip2Nx2NCandidatesInjection (:1608-1700).  Returns the new candidate count. */
MD_FN int md_inter_candidates(const SvtAmdMdPicture *P, const SvtAmdMdLcu *L, const SvtAmdMeCuResult *me, const MdInterLists *T, uint32_t ox,
                              uint32_t oy, int totalMerge, MdCand *cand, int n)
{
    for (int i = 0; i < me->total_me_candidate_index; i++) {
        const int dir = me->direction[i];
        if (dir == MD_BI && P->depth_mode == 0 && L->lcu_md_mode == 10)
            continue;
        MdCand *c = &cand[n];
        c->type = MD_INTER, c->intra_mode = 0, c->mpm = 0, c->dist_ready = 1, c->me_dist = me->distortion[i];
        c->dir = (uint8_t)dir, c->merge_flag = 0, c->merge_index = 0;
        c->mv[0].x = me->x_mv_l0, c->mv[0].y = me->y_mv_l0, c->mv[1].x = me->x_mv_l1, c->mv[1].y = me->y_mv_l1;
        c->mvp_idx[0] = c->mvp_idx[1] = 0, c->mvp[0].x = c->mvp[0].y = c->mvp[1].x = c->mvp[1].y = 0;
        md_choose_mvp(P, ox, oy, T, c);
        n++;
    }
    for (int k = 0; k < totalMerge; k++) {
        if (k >= T->merge_count)
            continue;
        const MdMergeCand *mc = &T->merge[k];
        int dup = 0;
        for (int j = k; j > 0 && !dup;) {
            const MdMergeCand *d = &T->merge[--j];
            const int f0 = mc->mv[0].x == d->mv[0].x && mc->mv[0].y == d->mv[0].y;
            const int f1 = mc->dir != MD_L0 && mc->mv[1].x == d->mv[1].x && mc->mv[1].y == d->mv[1].y;
            const int same = mc->dir == MD_L0 ? f0 : (mc->dir == MD_L1 ? f1 : (f0 && f1));
            dup = mc->dir == d->dir && same;
        }
        if (dup)
            continue;
        MdCand *c = &cand[n];
        c->type = MD_INTER, c->intra_mode = 0, c->mpm = 0, c->dist_ready = 0, c->me_dist = 0;
        c->dir = mc->dir, c->merge_flag = 1, c->merge_index = (uint8_t)k;
        c->mv[0] = mc->mv[0], c->mv[1] = mc->mv[1];
        c->mvp_idx[0] = c->mvp_idx[1] = 0, c->mvp[0].x = c->mvp[0].y = c->mvp[1].x = c->mvp[1].y = 0;
        n++;
    }
    return n;
}

/* mvBitTable (Codec/EbModeDecisionConfiguration.h:108): the 500 x 500 table is a 3 x 3 core plus 2 bits (1 << 16) per doubling of
 * either component beyond 2 (checked entry by entry against the header by tests/test_oracle_md_golden.py::test_mv_bit_table) */
/* entry i of a table of three constants: selects, not a load (on the device a table in memory is a round trip on the decision chain of every unit) */
MD_FN uint32_t md_sel3(int i, uint32_t a, uint32_t b, uint32_t c) { return i <= 0 ? a : (i == 1 ? b : c); }
MD_FN uint32_t md_mv_bits(int mvdX, int mvdY)
{
    const int row = mvdX > 2 ? 2 : mvdX, col = mvdY > 2 ? 2 : mvdY;
    const uint32_t core = md_sel3(row, md_sel3(col, 73744, 128728, 203592), md_sel3(col, 130975, 178780, 253644), md_sel3(col, 202683, 253623, 321933));
    /* doublings beyond 2: floor(log2 v) - 1 for v >= 4 */
    const int lx = mvdX >= 4 ? 30 - __builtin_clz((unsigned)mvdX) : 0, ly = mvdY >= 4 ? 30 - __builtin_clz((unsigned)mvdY) : 0;
    return core + 65536u * (uint32_t)(lx + ly);
}
MD_FN uint32_t md_mvd_rate(const MdCand *c, int list)
{
    int dx = c->mvp[list].x > c->mv[list].x ? c->mvp[list].x - c->mv[list].x : c->mv[list].x - c->mvp[list].x;
    int dy = c->mvp[list].y > c->mv[list].y ? c->mvp[list].y - c->mv[list].y : c->mv[list].y - c->mvp[list].y;
    dx = dx > 499 ? 499 : dx, dy = dy > 499 ? 499 : dy;
    return md_mv_bits(dx, dy) + (c->mvp_idx[list] ? 44891u : 23196u); /* mvpIndexBits, EbRateDistortionCost.c:18 */
}
/* getWeightedChromaDistortion (Codec/EbRateDistortionCost.c:35-65): chromaWeight = ChromaWeightFactor*[qp] of the picture's class (SvtAmdMdInter.chroma_weight) */
MD_FN uint64_t md_weighted_chroma(uint64_t chromaDistortion, uint32_t chromaWeight) { return (chromaDistortion * chromaWeight + 128) >> 8; }
/* InterFastCostPsliceOpt / InterFastCostBsliceOpt (Codec/EbRateDistortionCost.c:1157-1465).  chromaDistortion: the Cb + Cr SAD of a candidate of a
 * CHROMA_MODE_FULL LCU (0 otherwise); weightChroma: candidateBuffer->weightChromaDistortion (0 in LCUs of noise class CMPLX_NOISE,
 * Codec/EbProductCodingLoop.c:2079-2094: a merge candidate's chroma SAD is then added unweighted) */
MD_FN uint64_t md_inter_fast_cost_c(const SvtAmdMdPicture *P, const MdStats *st, const MdCu *cu, const MdCand *c, uint64_t lumaDistortion,
                                    uint64_t chromaDistortion, uint32_t chromaWeight, int weightChroma, uint64_t *fastLumaRate)
{
    /* skipFlagBits[3 + skip context] = {54723, 14816, 8254}; mergeIndexBits = {10350, 109741, 142509, 175277, 175277};
     * interBiDirBits[2 depth + bi] = {29856, 36028, 15752, 59703, 8692, 84420, 2742, 136034}; interUniDirBits = {2742, 136034} */
    uint64_t rate;
    if (c->merge_flag) {
        rate = (uint64_t)md_sel3(cu->skip_ctx, 54723, 14816, 8254) + (c->merge_index >= 3 ? 175277u : md_sel3(c->merge_index, 10350, 109741, 142509));
        *fastLumaRate = rate;
        const uint64_t distortion = weightChroma ? (lumaDistortion << 8) + md_weighted_chroma(chromaDistortion, chromaWeight) : (lumaDistortion + chromaDistortion) << 8;
        return distortion + (((uint64_t)P->fast_lambda * rate + (1u << 22)) >> 23);
    }
    rate = 86440;
    if (P->slice_type == 0) {
        const int bi = c->dir == MD_BI;
        rate += st->depth >= 3 ? (bi ? 136034u : 2742u) : md_sel3(st->depth, bi ? 36028u : 29856u, bi ? 59703u : 15752u, bi ? 84420u : 8692u);
        if (c->dir != MD_BI) /* (constant list indices: a run-time index into the candidate record would pin it to memory on the device) */
            rate += c->dir ? 136034u + md_mvd_rate(c, 1) : 2742u + md_mvd_rate(c, 0);
        else
            rate += (uint64_t)md_mvd_rate(c, 0) + md_mvd_rate(c, 1);
    } else {
        rate += md_mvd_rate(c, 0);
    }
    *fastLumaRate = rate;
    return (lumaDistortion << 8) + md_weighted_chroma(chromaDistortion, chromaWeight) + (((uint64_t)P->fast_lambda * rate + (1u << 22)) >> 23);
}
MD_FN uint64_t md_inter_fast_cost(const SvtAmdMdPicture *P, const MdStats *st, const MdCu *cu, const MdCand *c, uint64_t lumaDistortion,
                                  uint64_t *fastLumaRate)
{
    return md_inter_fast_cost_c(P, st, cu, c, lumaDistortion, 0, 0, 1, fastLumaRate);
}
/* Intra2Nx2NFastCostPsliceOpt (:580-660) */
MD_FN uint64_t md_intra_fast_cost_pslice_c(const SvtAmdMdPicture *P, const MdStats *st, const MdCu *c, int lumaMode, uint64_t lumaDistortion,
                                           uint64_t chromaDistortion, uint32_t chromaWeight, uint64_t *fastLumaRate)
{
    const uint64_t chromaRate = 12368;
    uint64_t lumaRate = st->depth == 3 ? 31523 : 0;
    lumaRate += 136034;
    lumaRate += (lumaMode == c->left_intra_mode || lumaMode == c->top_intra_mode) ? 72731 : 192228;
    *fastLumaRate = lumaRate;
    return (lumaDistortion << 8) + md_weighted_chroma(chromaDistortion, chromaWeight) + (((uint64_t)P->fast_lambda * (lumaRate + chromaRate) + (1u << 22)) >> 23);
}
MD_FN uint64_t md_intra_fast_cost_pslice(const SvtAmdMdPicture *P, const MdStats *st, const MdCu *c, int lumaMode, uint64_t lumaDistortion,
                                         uint64_t *fastLumaRate)
{
    return md_intra_fast_cost_pslice_c(P, st, c, lumaMode, lumaDistortion, 0, 0, fastLumaRate);
}
/* the noise-class rule of the fast loop's chroma distortion (Codec/EbProductCodingLoop.c:2079-2094): in an LCU of class CMPLX_NOISE the chroma SAD of a 64x64
 * candidate that does not move (intra candidates count: their vectors are zero) is quartered */
MD_FN uint64_t md_fast_chroma_noise_rule(const SvtAmdMdLcu *L, int cuSize, const MdCand *c, uint64_t chromaDistortion)
{
    if (!L->cmplx_noise || cuSize != 64)
        return chromaDistortion;
    const int l0zz = (c->dir & 1) ? 1 : (c->mv[0].x == 0 && c->mv[0].y == 0);
    const int l1zz = c->dir > 0 ? (c->mv[1].x == 0 && c->mv[1].y == 0) : 1;
    return (l0zz && l1zz) ? chromaDistortion >> 2 : chromaDistortion;
}

/* the rate of the transform-tree flags of a unit's luma (shared tail of InterFullLumaCost / MergeSkipFullLumaCost / IntraFullLumaCostPslice):
 * one split flag + one cbf per transform unit (four 32x32 units, ycbf bits 1..4, in a 64x64 unit) */
MD_FN uint64_t md_tu_flags_rate(const SvtAmdMdPicture *P, int cuSize, uint32_t ycbf)
{
    if (cuSize == 64) {
        uint64_t r = 0;
        for (int tu = 1; tu <= 4; tu++)
            r += (uint64_t)P->rates.transSubDivFlagBits[0] + P->rates.lumaCbfBits[((ycbf >> tu) & 1) * 5 + 0];
        return r;
    }
    const int lg = cuSize == 32 ? 5 : (cuSize == 16 ? 4 : 3);
    return (uint64_t)P->rates.transSubDivFlagBits[5 - lg] + P->rates.lumaCbfBits[(ycbf > 0) * 5 + 1];
}
/* InterFullLumaCost (:1968-2075) incl. MergeSkipFullLumaCost (:2370-2520).  dist[0] / dist[1]: yFullDistortion[DIST_CALC_RESIDUAL /
 * PREDICTION]; merge units also get their merge and skip costs. */
MD_FN uint64_t md_inter_full_luma_cost(const SvtAmdMdPicture *P, const MdCu *cu, const MdCand *c, int cuSize, uint32_t ycbf, uint64_t fastLumaRate,
                                       const uint64_t dist[2], uint64_t yCoeffBits, uint64_t *mergeCost, uint64_t *skipCost)
{
    const uint64_t lambda = P->full_lambda, coeffRate = yCoeffBits << 15;
    const int rootCbf = ycbf != 0;
    if (c->merge_flag) {
        uint64_t rate = (uint64_t)P->rates.skipFlagBits[cu->skip_ctx] + P->rates.mergeFlagBits[1] + P->rates.predModeBits[0] + P->rates.interPartSizeBits[0] +
                        P->rates.mergeIndexBits[c->merge_index];
        if (rootCbf)
            rate += md_tu_flags_rate(P, cuSize, ycbf);
        const uint64_t mc = (dist[0] << 8) + (((lambda * coeffRate + lambda * rate) + (1u << 22)) >> 23);
        const uint64_t sc = (dist[1] << 8) + (((lambda * fastLumaRate) + (1u << 22)) >> 23);
        *mergeCost = mc, *skipCost = sc;
        return sc <= mc ? sc : mc;
    }
    uint64_t rate = P->rates.rootCbfBits[rootCbf];
    if (rootCbf)
        rate += md_tu_flags_rate(P, cuSize, ycbf);
    rate += fastLumaRate;
    return (dist[0] << 8) + (((lambda * coeffRate + lambda * rate) + (1u << 22)) >> 23);
}
/* IntraFullLumaCostPslice (:1060-1140): units up to 32x32, one transform unit */
MD_FN uint64_t md_intra_full_luma_cost_pslice(const SvtAmdMdPicture *P, int cuSize, uint32_t ycbf, uint64_t fastLumaRate, uint64_t yDistortion0,
                                              uint64_t yCoeffBits)
{
    const uint64_t lambda = P->full_lambda, coeffRate = yCoeffBits << 15;
    const int lg = cuSize == 32 ? 5 : (cuSize == 16 ? 4 : 3);
    const uint64_t rate = (uint64_t)P->rates.transSubDivFlagBits[5 - lg] + P->rates.lumaCbfBits[(ycbf & 1) * 5 + 1] + fastLumaRate;
    return (yDistortion0 << 8) + (((lambda * coeffRate + lambda * rate) + (1u << 22)) >> 23);
}
/* the scaling PerformFullLoop applies to the luma coefficient bits of partial-frequency units (:4575-4590; N2_TH by QP) */
MD_FN uint64_t md_pf_coeff_bits(int pfMode, int qp, uint64_t yCoeffBits)
{
    if (pfMode == 1)
        return yCoeffBits * (uint64_t)(qp < 10 ? 3 : (qp < 29 ? 2 : 1));
    return yCoeffBits;
}
/* DerivePartialFrequencyN2Flag (:2243-2260), levels 0 / 1 */
MD_FN int md_pf_mode(const SvtAmdMdPicture *P) { return P->pf_md_level == 1 ? 1 : 0; }

/* SkipSmallCu (:2292-2306) and its caller's condition (:4826-4836): returns the unit's split flag */
MD_FN int md_skip_small_cu(const SvtAmdMdPicture *P, const SvtAmdMdLcu *L, const MdLcuState *S, int leaf, int depth)
{
    const int applies = (P->depth_mode >= 5 || (P->depth_mode == 0 && (L->lcu_md_mode == 5 || L->lcu_md_mode == 6 || L->lcu_md_mode == 7))) && L->is_complete;
    if (applies && L->skip_small_cu && S->local[leaf].left_depth == depth && S->local[leaf].top_depth == depth && S->cu[leaf].split)
        return 0;
    return S->cu[leaf].split;
}

/* StopSplitCondition (Codec/EbFullLoop.c:1392-1455): depth thresholds on the unit's luma distortion; the tables (:14-70) depend on the
 * temporal layer only (layer <= hierarchical levels) */
/* ... as a threshold of the unit's depth that the LCU fixes (0: the condition never holds): the device derives the three once per LCU */
MD_FN uint32_t md_stop_split_threshold(const SvtAmdMdPicture *P, const SvtAmdMdLcu *L, int depth)
{
    const uint16_t d0[2][6] = {{1000, 4000, 9500, 3000, 3000, 3000}, {0, 1000, 7000, 9500, 9500, 9500}};
    const uint16_t d1[2][6] = {{0, 2000, 5500, 9500, 9500, 9500}, {0, 1500, 1500, 1500, 1500, 1500}};
    const uint16_t d2[2][6] = {{0, 500, 2000, 2500, 2500, 2500}, {0, 1500, 1000, 4500, 4500, 4500}};
    if (P->depth_mode == 1 || P->depth_mode == 2 || (P->depth_mode == 0 && (L->lcu_md_mode == 1 || L->lcu_md_mode == 2 || L->lcu_md_mode == 7)))
        return 0;
    if (P->temporal_layer == 0 || P->slice_type == 2)
        return 0;
    if (!L->is_complete || L->no_stop_split)
        return 0;
    const int e = L->edge_block != 0, t = P->temporal_layer > 5 ? 5 : P->temporal_layer;
    return depth == 0 ? d0[e][t] : depth == 1 ? d1[e][t] : depth == 2 ? d2[e][t] : 0u;
}
MD_FN int md_stop_split(const SvtAmdMdPicture *P, const SvtAmdMdLcu *L, int depth, uint32_t fullDistortion)
{
    return fullDistortion < md_stop_split_threshold(P, L, depth);
}

/* the transform-tree flag rates of a unit with chroma (shared by InterFullCost :1753-1950 and MergeSkipFullCost :2107-2355; rootCbf != 0): the luma part
 * (split flags + luma cbf flags) and the chroma cbf flags.  yCbf / cbCbf / crCbf: bit 0 of a unit below 64x64, bits 1..4 of the four 32x32 transform units of a 64x64 unit */
MD_FN void md_tu_flags_rate_c(const SvtAmdMdPicture *P, int cuSize, uint32_t yCbf, uint32_t cbCbf, uint32_t crCbf, uint64_t *lumaFlags, uint64_t *chromaFlags)
{
    const int lgT = cuSize == 64 ? 5 : (cuSize == 32 ? 5 : (cuSize == 16 ? 4 : 3));
    uint64_t lf = 0, cf = 0;
    if (cuSize == 64) {
        cf += (uint64_t)P->rates.chromaCbfBits[(crCbf > 0) * 5 + 0] + P->rates.chromaCbfBits[(cbCbf > 0) * 5 + 0];
        for (int tu = 1; tu <= 4; tu++) {
            lf += (uint64_t)P->rates.transSubDivFlagBits[5 - lgT] + P->rates.lumaCbfBits[((yCbf >> tu) & 1) * 5 + 0];
            cf += crCbf > 0 ? P->rates.chromaCbfBits[((crCbf >> tu) & 1) * 5 + 1] : 0;
            cf += cbCbf > 0 ? P->rates.chromaCbfBits[((cbCbf >> tu) & 1) * 5 + 1] : 0;
        }
    } else {
        lf += P->rates.transSubDivFlagBits[5 - lgT];
        if (cbCbf > 0 || crCbf > 0)
            lf += P->rates.lumaCbfBits[(yCbf > 0) * 5 + 1];
        cf += (uint64_t)P->rates.chromaCbfBits[(cbCbf > 0) * 5 + 0] + P->rates.chromaCbfBits[(crCbf > 0) * 5 + 0];
    }
    *lumaFlags = lf, *chromaFlags = cf;
}
/* MergeSkipFullCost (Codec/EbRateDistortionCost.c:2107-2355): the merge and the skip cost of a merge candidate from the sums of its luma and chroma full loops.
 * cbf[p] / bits[p] / dist[p][2]: candidatePtr->cbCbf / crCbf, *cbCoeffBits / *crCoeffBits, cb / crFullDistortion. */
MD_FN void md_merge_skip_full_cost_v(const SvtAmdMdPicture *P, uint32_t chromaWeight, int skipCtx, int mergeIndex, int cuSize, uint32_t yCbf, const uint32_t cbf[2],
                                     uint64_t yCoeffBits, const uint64_t bits[2], const uint64_t yDist[2], const uint64_t dist[2][2], uint64_t fastLumaRate,
                                     uint64_t *mergeCost, uint64_t *skipCost)
{
    const uint32_t cbCbf = cbf[0], crCbf = cbf[1];
    const int rootCbf = yCbf || cbCbf || crCbf;
    uint64_t lumaFlags = 0, chromaFlags = 0;
    if (rootCbf)
        md_tu_flags_rate_c(P, cuSize, yCbf, cbCbf, crCbf, &lumaFlags, &chromaFlags);
    const uint64_t mergeLumaRate = (uint64_t)P->rates.skipFlagBits[skipCtx] + P->rates.mergeFlagBits[1] + P->rates.predModeBits[0] +
                                   P->rates.interPartSizeBits[0] + P->rates.mergeIndexBits[mergeIndex] + lumaFlags;
    const uint64_t coeffRate = (yCoeffBits + bits[0] + bits[1]) << 15;
    const uint64_t lambda = P->full_lambda, lambdaChroma = P->full_chroma_lambda;
    const uint64_t mergeChroma = md_weighted_chroma(dist[0][0] + dist[1][0], chromaWeight), skipChroma = md_weighted_chroma(dist[0][1] + dist[1][1], chromaWeight);
    *mergeCost = (yDist[0] << 8) + mergeChroma + (((lambda * coeffRate + lambda * mergeLumaRate + lambdaChroma * chromaFlags) + (1u << 22)) >> 23);
    *skipCost = (yDist[1] << 8) + skipChroma + (((lambda * fastLumaRate) + (1u << 22)) >> 23);
}
/* ... as AddChromaEncDec calls it for the merge unit the mode decision chose in a CHROMA_MODE_BEST LCU (Codec/EbProductCodingLoop.c:4158-4349): the luma terms
 * the mode decision kept (MdCu) + the chroma loop's sums */
MD_FN void md_merge_skip_full_cost(const SvtAmdMdPicture *P, const SvtAmdMdInter *X, const MdCu *cu, int cuSize, const uint32_t cbf[2], const uint64_t bits[2],
                                   const uint64_t dist[2][2], uint64_t *mergeCost, uint64_t *skipCost)
{
    md_merge_skip_full_cost_v(P, X->chroma_weight, cu->skip_ctx, cu->merge_index, cuSize, cu->ycbf_mask, cbf, cu->y_coeff_bits, bits, cu->y_dist, dist, cu->fast_luma_rate,
                              mergeCost, skipCost);
}
/* InterFullCost (:1753-1950) of a candidate of a CHROMA_MODE_FULL LCU; merge candidates: MergeSkipFullCost (full cost = the smaller of the two) */
MD_FN uint64_t md_inter_full_cost(const SvtAmdMdPicture *P, uint32_t chromaWeight, const MdCu *cu, const MdCand *c, int cuSize, uint32_t yCbf, const uint32_t cbf[2],
                                  uint64_t fastLumaRate, const uint64_t yDist[2], const uint64_t dist[2][2], uint64_t yCoeffBits, const uint64_t bits[2],
                                  uint64_t *mergeCost, uint64_t *skipCost)
{
    if (c->merge_flag) {
        md_merge_skip_full_cost_v(P, chromaWeight, cu->skip_ctx, c->merge_index, cuSize, yCbf, cbf, yCoeffBits, bits, yDist, dist, fastLumaRate, mergeCost, skipCost);
        return *skipCost <= *mergeCost ? *skipCost : *mergeCost;
    }
    const int rootCbf = yCbf || cbf[0] || cbf[1];
    uint64_t lumaFlags = 0, chromaFlags = 0;
    if (rootCbf)
        md_tu_flags_rate_c(P, cuSize, yCbf, cbf[0], cbf[1], &lumaFlags, &chromaFlags);
    const uint64_t lumaRate = (uint64_t)P->rates.rootCbfBits[rootCbf] + lumaFlags + fastLumaRate;
    const uint64_t coeffRate = (yCoeffBits + bits[0] + bits[1]) << 15, lambda = P->full_lambda, lambdaChroma = P->full_chroma_lambda;
    return (yDist[0] << 8) + md_weighted_chroma(dist[0][0] + dist[1][0], chromaWeight) + (((lambda * coeffRate + lambda * lumaRate + lambdaChroma * chromaFlags) + (1u << 22)) >> 23);
}
/* IntraFullCostPslice (:807-950) of a candidate of a CHROMA_MODE_FULL LCU: units up to 32x32, one transform unit; fastChromaRate = 12368 */
MD_FN uint64_t md_intra_full_cost_pslice(const SvtAmdMdPicture *P, uint32_t chromaWeight, int cuSize, uint32_t yCbf, const uint32_t cbf[2], uint64_t fastLumaRate,
                                         uint64_t yDistortion0, const uint64_t dist[2][2], uint64_t yCoeffBits, const uint64_t bits[2])
{
    const int lg = cuSize == 32 ? 5 : (cuSize == 16 ? 4 : 3);
    const uint64_t lumaRate = (uint64_t)P->rates.transSubDivFlagBits[5 - lg] + P->rates.lumaCbfBits[(yCbf & 1) * 5 + 1] + fastLumaRate;
    const uint64_t chromaRate = (uint64_t)P->rates.chromaCbfBits[(cbf[1] & 1) * 5 + 0] + P->rates.chromaCbfBits[(cbf[0] & 1) * 5 + 0] + 12368;
    const uint64_t coeffRate = (yCoeffBits + bits[0] + bits[1]) << 15, lambda = P->full_lambda, lambdaChroma = P->full_chroma_lambda;
    return (yDistortion0 << 8) + md_weighted_chroma(dist[0][0] + dist[1][0], chromaWeight) + (((lambda * coeffRate + lambda * lumaRate + lambdaChroma * chromaRate) + (1u << 22)) >> 23);
}
/* the merge / skip decision of EncodePass for a merge unit (Codec/EbCodingLoop.c:3838-3882): 2 = SVT_AMD_EP_INTER_SKIP, 1 = _MERGE */
MD_FN int md_ep_merge_kind(const SvtAmdMdInter *X, const SvtAmdMdLcu *L, uint64_t mergeCost, uint64_t skipCost)
{
    if (X->skip_cost_bias && L->variance_below_200)
        skipCost += (skipCost * 70) / 100;
    return skipCost <= mergeCost ? 2 : 1;
}

/* what this revision of the device call covers (include/svt_hevc_amd.h) */
MD_FN int md_picture_supported(const SvtAmdMdPicture *P)
{
    return P->slice_type == 2 && P->depth_mode == 2 /* PICT_FULL84 */ && !P->intra_md_open_loop && P->chroma_level == 1 && !P->coeff_cabac_update &&
           P->intra4x4_level == 2 && !P->rdoq_pmcore_method && !P->single_fast_loop && !P->spatial_sse_full_loop && P->pf_md_level == 0 &&
           P->nfl_level_md != 3 && P->intra_injection_method <= 2 && !(P->width & 7) && !(P->height & 7);
}
/* P / B pictures; the LCUs must in addition all be decided by ModeDecisionLcu (md_lcu_supported) */
MD_FN int md_picture_supported_inter(const SvtAmdMdPicture *P, const SvtAmdMdInter *X)
{
    return P->slice_type != 2 && P->intra_md_open_loop && !P->coeff_cabac_update && P->intra4x4_level == 2 &&
           !P->rdoq_pmcore_method && !P->single_fast_loop && !P->spatial_sse_full_loop && P->pf_md_level <= 1 && P->nfl_level_md != 3 &&
           P->intra_injection_method <= 1 && !P->limit_ois_to_dc_mode && !P->mpm_search && P->enc_mode < 10 && !(P->width & 7) && !(P->height & 7) &&
           X->use_subpel && X->unrestricted_mv && X->generate_amvp_table_md && !X->extra_injection && !X->improve_sharpness;
}
MD_FN int md_lcu_supported(const SvtAmdMdPicture *P, const SvtAmdMdLcu *L)
{
    if (L->chroma_encode_mode != 2 /* CHROMA_MODE_BEST */ && L->chroma_encode_mode != 1 /* CHROMA_MODE_FULL: chroma in both loops of every candidate */)
        return 0;
    if (P->depth_mode == 0) /* PICT_LCU_SWITCH: branch-and-depth-pillar LCUs (3, 4) stay with the reference */
        return L->lcu_md_mode != 3 && L->lcu_md_mode != 4 && L->lcu_md_mode != 0;
    return P->depth_mode == 1 || P->depth_mode == 2 || P->depth_mode == 5;
}
#endif
