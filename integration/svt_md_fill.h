/*
 * integration/svt_md_fill.h - the reference-side half of the mode-decision input contract (include/svt_hevc_amd.h "Device-resident
 * mode decision"): reads the controls the reference's host code has derived for a picture / an LCU out of its own structures.
 * Shared by the binding (integration/svt_hook_md.c) and by the recording harness of the test fixtures (oracle/ref_harness_md_dump.c),
 * so that fixtures hold exactly what the binding hands to the device.  Part of the code a maintainer adds to the reference; contains
 * no reference source.
 *
 * svt_md_fill_picture: call after ModeDecisionConfigureLcu (Codec/EbEncDecProcess.c:2893) has run for ANY LCU of the picture on `md` (the
 * fields it reads are the same for every LCU of a picture without the delta-QP tools).  svt_md_fill_lcu reads only picture-level state,
 * so the binding can fill every LCU of a picture at the picture's first ModeDecisionLcu call.
 */
#ifndef SVT_MD_FILL_H
#define SVT_MD_FILL_H
#include <string.h>
#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbModeDecisionProcess.h"
#include "EbModeDecisionConfiguration.h"
#include "EbMdRateEstimation.h"
#include "../include/svt_hevc_amd.h"

_Static_assert(sizeof(MdRateEstimationContext_t) == sizeof(SvtAmdMdRates), "MdRateEstimationContext_t layout");
_Static_assert(offsetof(MdRateEstimationContext_t, intraLumaBits) == offsetof(SvtAmdMdRates, intraLumaBits), "intraLumaBits");
_Static_assert(offsetof(MdRateEstimationContext_t, lumaCbfBits) == offsetof(SvtAmdMdRates, lumaCbfBits), "lumaCbfBits");
_Static_assert(offsetof(MdRateEstimationContext_t, transSubDivFlagBits) == offsetof(SvtAmdMdRates, transSubDivFlagBits), "transSubDivFlagBits");
_Static_assert(offsetof(MdRateEstimationContext_t, interUniDirBits) == offsetof(SvtAmdMdRates, interUniDirBits), "interUniDirBits");

EB_U8 DeriveContouringClass(PictureParentControlSet_t *parentPcsPtr, EB_U16 lcuIndex, EB_U8 leafIndex); /* EbModeDecisionConfiguration.c:395 */

static void svt_md_fill_picture(SvtAmdMdPicture *P, const SequenceControlSet_t *scs, const PictureControlSet_t *pcs, const ModeDecisionContext_t *md)
{
    const PictureParentControlSet_t *pp = pcs->ParentPcsPtr;
    memset(P, 0, sizeof(*P));
    P->width = (uint16_t)scs->lumaWidth, P->height = (uint16_t)scs->lumaHeight;
    P->slice_type = (uint8_t)pcs->sliceType, P->temporal_layer = pcs->temporalLayerIndex, P->is_reference = pp->isUsedAsReferenceFlag;
    P->enc_mode = pcs->encMode, P->depth_mode = (uint8_t)pp->depthMode;
    P->intra_md_open_loop = md->intraMdOpenLoopFlag, P->intra_injection_method = md->intraInjectionMethod, P->limit_intra = md->limitIntra;
    P->mpm_search = md->mpmSearch, P->mpm_search_candidate = md->mpmSearch ? md->mpmSearchCandidate : 0;
    P->pf_md_level = md->pfMdLevel, P->nfl_level_md = md->nflLevelMd, P->nmm_level_md = md->nmmLevelMd;
    P->full_loop_escape = md->fullLoopEscape, P->single_fast_loop = md->singleFastLoopFlag, P->coeff_cabac_update = md->coeffCabacUpdate;
    P->spatial_sse_full_loop = md->spatialSseFullLoop, P->chroma_level = md->chromaLevel, P->intra4x4_level = md->intra4x4Level;
    P->rdoq_pmcore_method = (uint8_t)md->rdoqPmCoreMethod;
    P->skip_ois_8x8 = pp->skipOis8x8, P->cu8x8_mode = pp->cu8x8Mode, P->cu16x16_mode = pp->cu16x16Mode, P->limit_ois_to_dc_mode = pp->limitOisToDcModeFlag;
    P->constrained_intra = pcs->constrainedIntraFlag, P->strong_smoothing = scs->enableStrongIntraSmoothing;
    P->qp = md->qp, P->chroma_qp = md->chromaQp;
    P->fast_lambda = md->fastLambda, P->full_lambda = md->fullLambda, P->fast_chroma_lambda = md->fastChromaLambda, P->full_chroma_lambda = md->fullChromaLambda;
    memcpy(&P->rates, md->mdRateEstimationPtr, sizeof(P->rates));
}

static void svt_md_fill_lcu(SvtAmdMdLcu *L, const SequenceControlSet_t *scs, const PictureControlSet_t *pcs, const LargestCodingUnit_t *lcuPtr,
                            const ModeDecisionContext_t *md)
{
    PictureParentControlSet_t *pp = pcs->ParentPcsPtr;
    const EB_U32 lcu = lcuPtr->index;
    const MdcLcuData_t *mdc = &pcs->mdcLcuArray[lcu];
    memset(L, 0, sizeof(*L));
    L->leaf_count = (uint8_t)mdc->leafCount;
    for (EB_U32 i = 0; i < mdc->leafCount && i < SVT_AMD_MD_LEAVES; i++)
        L->leaf_index[i] = mdc->leafDataArray[i].leafIndex, L->leaf_split[i] = mdc->leafDataArray[i].splitFlag;
    L->tile_left = lcuPtr->lcuEdgeInfoPtr->tileLeftEdgeFlag, L->tile_top = lcuPtr->lcuEdgeInfoPtr->tileTopEdgeFlag;
    L->tile_right = lcuPtr->lcuEdgeInfoPtr->tileRightEdgeFlag;
    L->is_complete = scs->lcuParamsArray[lcu].isCompleteLcu;
    L->complexity_status_2 = pp->complexLcuArray[lcu] == LCU_COMPLEXITY_STATUS_2;
    for (int q = 0; q < 4; q++)
        L->contouring_class[q] = DeriveContouringClass(pp, (EB_U16)lcu, (EB_U8)(1 + 21 * q));
    /* ConfigureChroma (Codec/EbModeDecisionProcess.c:408-471) for the two levels that need no per-LCU detector; 0 = a switch level (2..5) */
    L->chroma_encode_mode = md->chromaLevel == 0 && pcs->colorFormat < EB_YUV422 ? CHROMA_MODE_FULL : md->chromaLevel <= 1 ? CHROMA_MODE_BEST : 0;
    /* contextPtr->mdContext->restrictIntraGlobalMotion (Codec/EbEncDecProcess.c:2890) */
    L->restrict_intra_global_motion = (pp->isPan || pp->isTilt) && pp->nonMovingIndexArray[lcu] < INTRA_GLOBAL_MOTION_NON_MOVING_INDEX_TH &&
                                      pp->yMean[lcu][RASTER_SCAN_CU_INDEX_64x64] < INTRA_GLOBAL_MOTION_DARK_LCU_TH;
    L->lcu_md_mode = pp->depthMode == PICT_LCU_SWITCH_DEPTH_MODE ? pp->lcuMdModeArray[lcu] : 0;
}

/* the picture's open-loop intra search results in the contract's layout (SvtAmdOisLcuResult: by raster-scan CU index) */
static void svt_md_fill_ois(SvtAmdOisLcuResult *o, const PictureParentControlSet_t *pp, EB_U32 lcu)
{
    memset(o, 0, sizeof(*o));
    const OisCu32Cu16Results_t *a = pp->oisCu32Cu16Results[lcu];
    const OisCu8Results_t *b = pp->oisCu8Results[lcu];
    for (int cu = 1; cu < SVT_AMD_ME_PU_COUNT; cu++) {
        const OisCandidate_t *c = cu < 21 ? a->sortedOisCandidate[cu] : b->sortedOisCandidate[cu - 21];
        o->total_intra_luma_mode[cu] = cu < 21 ? a->totalIntraLumaMode[cu] : b->totalIntraLumaMode[cu - 21];
        for (int k = 0; k < SVT_AMD_OIS_MAX_CAND && k < MAX_OIS_2; k++)
            o->candidate[cu][k] = c[k].oisResults;
    }
}
#endif
