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
    uint64_t cost;
} MdLocal;
typedef struct MdCu {
    uint8_t split, pred_mode, intra_luma_mode, ycbf, skip_flag;
    uint8_t left_intra_mode, top_intra_mode; /* PredictionUnit_t.intraLumaLeftMode / TopMode */
    uint8_t skip_ctx;
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
    if (P->slice_type != 2)
        return 0; /* P / B pictures: not in this revision */
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
    int evaluated_count;      /* secondFastCostSearchCandidateTotalCount */
} MdBuffers;
MD_FN void md_fast_loop_buffers(MdBuffers *B, int width, int maxBuffers, int ncand, const uint64_t *costs, const uint8_t *evaluated)
{
    for (int i = 0; i < MD_MAX_BUF; i++)
        B->fast_cost[i] = B->full_cost[i] = ~0ull, B->cand[i] = -1; /* EbHevcProductCodingLoopInitFastLoop :1586-1596 (i < width) */
    (void)width;
    B->evaluated_count = 0;
    int highest = 0;
    for (int idx = ncand - 1; idx >= 0; idx--) {
        B->cand[highest] = (int16_t)idx;
        if (evaluated[idx]) {
            B->fast_cost[highest] = costs[idx];
            B->evaluated_count++;
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

/* ProductPerformInterDepthDecision (Codec/EbFullLoop.c:1461-1776; StopSplitCondition :1382 is false in the depth modes of this revision)
 * and, with exit_parent != 0, ExitInterDepthDecision (:1070-1378).  Returns lastCuIndex. */
MD_FN int md_inter_depth_decision(const SvtAmdMdPicture *P, MdLcuState *S, int leaf, int lcu_x, int lcu_y, int exit_parent)
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
    } else if (S->cu[leaf].split == 0) { /* lastDepthFlag */
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

/* what this revision of the device call covers (include/svt_hevc_amd.h) */
MD_FN int md_picture_supported(const SvtAmdMdPicture *P)
{
    return P->slice_type == 2 && P->depth_mode == 2 /* PICT_FULL84 */ && !P->intra_md_open_loop && P->chroma_level == 1 && !P->coeff_cabac_update &&
           P->intra4x4_level == 2 && !P->rdoq_pmcore_method && !P->single_fast_loop && !P->spatial_sse_full_loop && P->pf_md_level == 0 &&
           P->nfl_level_md != 3 && P->intra_injection_method <= 2 && !(P->width & 7) && !(P->height & 7);
}
#endif
