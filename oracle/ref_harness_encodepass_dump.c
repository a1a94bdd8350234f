/*
 * TEST INFRASTRUCTURE ONLY (oracle/): link-time interposer around the REFERENCE's EncodePass (Codec/EbCodingLoop.c:2989),
 * compiled only into oracle/_ref/libsvtref.so with -Wl,--wrap=EncodePass.
 *
 * When SVT_REF_ENCODEPASS_DUMP names a file, every call on an LCU whose coding units are all intra 2Nx2N units of 8..32 or inter
 * 2Nx2N units of 8..64 with the plain encode loop (4:2:0; 8-bit encodes in EpRecord, 10-bit encodes - EncodePass with is16bit - in
 * EpRecord16 with 16-bit source and reconstruction) is recorded, and before the first such LCU of a P / B picture the picture-level
 * inputs of the inter branch: the reference pictures it predicts from (whole padded buffers, EpRefRecord, once per reference
 * picture) and the coefficient-rate tables (pictureControlSetPtr->cabacCost, EpCostRecord).  Per LCU: BEFORE the call the final coding-unit list of the LCU and its source samples, in the layout of the product's
 * encode-pass input contract (SvtAmdLcuWork, include/svt_hevc_amd.h); AFTER it what the reference produced, in the layout of the
 * output contract (SvtAmdLcuResult): TransformUnit_t cbf / isOnlyDc / nzCoefCount, LargestCodingUnit_t.quantizedCoeff and the
 * LCU of the reconstruction buffer (run the encoder with the loop filters off, `-dlf 1 -sao 0`, so that the buffer still holds
 * the un-deblocked samples when the call returns).  tests/golden/make_encodepass_golden.py turns the dump into fixtures.
 *
 * Contains no reference source; includes the reference headers only to read its structs.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbEncDecProcess.h"
#include "EbReferenceObject.h"
#include "EbCodingUnit.h"
#include "EbTransformUnit.h"
#include "EbTransforms.h"
#include "EbUtility.h"
#include "EbAvailability.h"

#include "../include/svt_hevc_amd.h"

void __real_EncodePass(SequenceControlSet_t *scs, PictureControlSet_t *pcs, LargestCodingUnit_t *lcuPtr, EB_U32 tbAddr, EB_U32 lcuOriginX,
                       EB_U32 lcuOriginY, EB_U32 lcuQp, EB_BOOL enableSaoFlag, EncDecContext_t *contextPtr);

#define EP_DUMP_MAGIC 0x53415045U /* "EPAS" */
#define EP_REF_MAGIC 0x46525045U  /* "EPRF" */
#define EP_COST_MAGIC 0x43435045U /* "EPCC" */
typedef struct EpRecord {
    uint32_t magic, record_size;
    uint64_t picture_number;
    uint32_t width, height, lcu_index, dlf_off;
    uint64_t ref_poc[2]; /* reference pictures of list 0 / 1 (EpRefRecord.poc), ~0 = none */
    SvtAmdLcuWork work;
    SvtAmdLcuResult result;
} EpRecord;
typedef struct EpRecord16 { /* same head; record_size tells them apart */
    uint32_t magic, record_size;
    uint64_t picture_number;
    uint32_t width, height, lcu_index, dlf_off;
    uint64_t ref_poc[2];
    SvtAmdLcuWork16 work;
    SvtAmdLcuResult16 result;
} EpRecord16;
typedef struct EpRefRecord { /* followed by the three padded planes: strideY * (height + 2 originY) luma samples, then Cb, Cr */
    uint32_t magic, record_size;
    uint64_t poc;
    uint32_t bps, strideY, strideC, originX, originY, width, height, pad;
} EpRefRecord;
typedef struct EpCostRecord {
    uint32_t magic, record_size;
    uint64_t picture_number;
    SvtAmdCabacCost cost;
} EpCostRecord;

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static FILE *g_file;
static int g_state;
static uint64_t g_ref_done[64], g_cost_done[64];
static int g_nref, g_ncost;

_Static_assert(sizeof(SvtAmdCabacCost) == sizeof(CabacCost_t), "CabacCost_t layout");

/* once per reference picture / per picture: the picture-level inputs of the inter branch */
static void dump_inter_inputs(const PictureControlSet_t *pcs, int is16bit, uint64_t ref_poc[2])
{
    ref_poc[0] = ref_poc[1] = ~0ull;
    pthread_mutex_lock(&g_lock);
    for (int l = 0; l < (pcs->sliceType == EB_B_PICTURE ? 2 : pcs->sliceType == EB_P_PICTURE ? 1 : 0); l++) {
        const EbReferenceObject_t *ro = (const EbReferenceObject_t *)pcs->refPicPtrArray[l]->objectPtr;
        const EbPictureBufferDesc_t *b = is16bit ? ro->referencePicture16bit : ro->referencePicture;
        ref_poc[l] = ro->refPOC;
        int seen = 0;
        for (int i = 0; i < g_nref; i++)
            seen |= g_ref_done[i] == ro->refPOC;
        if (seen || g_nref >= 64)
            continue;
        g_ref_done[g_nref++] = ro->refPOC;
        const size_t bps = is16bit ? 2 : 1, rowsY = b->height + 2u * b->originY, rowsC = rowsY / 2;
        EpRefRecord h = {EP_REF_MAGIC, 0, ro->refPOC, (uint32_t)bps, b->strideY, b->strideCb, b->originX, b->originY, b->width, b->height, 0};
        h.record_size = (uint32_t)(sizeof(h) + bps * ((size_t)b->strideY * rowsY + 2u * (size_t)b->strideCb * rowsC));
        fwrite(&h, sizeof(h), 1, g_file);
        fwrite(b->bufferY, bps, (size_t)b->strideY * rowsY, g_file);
        fwrite(b->bufferCb, bps, (size_t)b->strideCb * rowsC, g_file);
        fwrite(b->bufferCr, bps, (size_t)b->strideCr * rowsC, g_file);
    }
    int seen = 0;
    for (int i = 0; i < g_ncost; i++)
        seen |= g_cost_done[i] == pcs->pictureNumber;
    if (!seen && g_ncost < 64) {
        g_cost_done[g_ncost++] = pcs->pictureNumber;
        EpCostRecord c;
        c.magic = EP_COST_MAGIC, c.record_size = (uint32_t)sizeof(c), c.picture_number = pcs->pictureNumber;
        memcpy(&c.cost, pcs->cabacCost, sizeof(c.cost));
        fwrite(&c, sizeof(c), 1, g_file);
    }
    fflush(g_file);
    pthread_mutex_unlock(&g_lock);
}

/* the coded leaves of the LCU in order; returns 0 when a unit is outside what the record (and the product path) covers */
static int fill_work(SvtAmdLcuWork *w, const SequenceControlSet_t *scs, const PictureControlSet_t *pcs, const LargestCodingUnit_t *lcuPtr,
                     EB_U32 lcuOriginX, EB_U32 lcuOriginY, const EncDecContext_t *contextPtr)
{
    /* the encode loop this record (and the product path) covers: no delta-QP segments, no forced cbf, no coefficient shaping
     * (EbCodingLoop.c:3161, :2377, EbEncDecProcess.c:2211) */
    const EB_BOOL useDeltaQp = (EB_BOOL)(scs->staticConfig.improveSharpness || scs->staticConfig.bitRateReduction || scs->staticConfig.segmentOvEnabled);
    /* CHROMA_MODE_BEST LCUs: EncodePass itself completes the merge / skip costs of their merge units with chroma (AddChromaEncDec, :3840-3863); their
     * inter_kind is read from the flags the pass leaves (fill_after) */
    const int inter_ok = !useDeltaQp && !contextPtr->fastEl;
    if (contextPtr->mdContext->rdoqPmCoreMethod != EB_NO_RDOQ && contextPtr->mdContext->rdoqPmCoreMethod != EB_PMCORE)
        return 0; /* RDOQ (encMode 0) */
    memset(w, 0, sizeof(*w));
    w->lcu_x = (uint16_t)lcuOriginX, w->lcu_y = (uint16_t)lcuOriginY;
    w->slice_type = (uint8_t)pcs->sliceType, w->temporal_layer = pcs->temporalLayerIndex;
    w->constrained_intra = pcs->constrainedIntraFlag, w->strong_smoothing = scs->enableStrongIntraSmoothing;
    w->tile_left = lcuPtr->lcuEdgeInfoPtr->tileLeftEdgeFlag, w->tile_top = lcuPtr->lcuEdgeInfoPtr->tileTopEdgeFlag;
    w->tile_right = lcuPtr->lcuEdgeInfoPtr->tileRightEdgeFlag;
    w->pm_core = contextPtr->mdContext->rdoqPmCoreMethod == EB_PMCORE; /* encMode 1..4: DecoupledQuantizeInvQuantizeLoops (EbTransforms.c:3009-3052) */
    EB_U32 cuItr = 0, n = 0;
    while (cuItr < CU_MAX_COUNT) {
        const CodingUnit_t *cu = lcuPtr->codedLeafArrayPtr[cuItr];
        if (cu->splitFlag) {
            cuItr++;
            continue;
        }
        const CodedUnitStats_t *st = GetCodedUnitStats(cuItr);
        const int intra = cu->predictionModeFlag == INTRA_MODE;
        if (n >= SVT_AMD_LCU_MAX_CUS || (intra && (cu->predictionUnitArray->intraLumaMode == EB_INTRA_MODE_4x4 || st->size > 32)) || (!intra && !inter_ok))
            return 0;
        SvtAmdLcuCu *u = &w->cu[n++];
        u->x = st->originX, u->y = st->originY, u->size = st->size, u->pred_mode = (uint8_t)cu->predictionModeFlag;
        u->intra_luma_mode = (uint8_t)cu->predictionUnitArray->intraLumaMode;
        if (!intra) {
            const PredictionUnit_t *pu = cu->predictionUnitArray;
            u->intra_luma_mode = 0;
            u->inter_dir = (uint8_t)pu->interPredDirectionIndex;
            for (int l = 0; l < 2; l++)
                u->mv[l][0] = pu->mv[l].x, u->mv[l][1] = pu->mv[l].y;
            /* merge / skip decision of the unit (EbCodingLoop.c:3838-3882) */
            u->inter_kind = SVT_AMD_EP_INTER_AMVP;
            if (pu->mergeFlag) {
                EB_U64 skipCost = contextPtr->mdContext->mdEpPipeLcu[cu->leafIndex].skipCost;
                if (pcs->sliceType == EB_B_PICTURE && pcs->ParentPcsPtr->isUsedAsReferenceFlag == EB_FALSE) {
                    static const EB_U8 INTRA_AREA_TH[MAX_TEMPORAL_LAYERS] = {40, 30, 30, 0, 0, 0};
                    const EbReferenceObject_t *r0 = (const EbReferenceObject_t *)pcs->refPicPtrArray[REF_LIST_0]->objectPtr;
                    const EbReferenceObject_t *r1 = (const EbReferenceObject_t *)pcs->refPicPtrArray[REF_LIST_1]->objectPtr;
                    if (pcs->ParentPcsPtr->variance[lcuPtr->index][0] < 200 &&
                        (r0->intraCodedArea > INTRA_AREA_TH[r0->tmpLayerIdx] || r1->intraCodedArea > INTRA_AREA_TH[r1->tmpLayerIdx]))
                        skipCost += (skipCost * 70) / 100;
                }
                u->inter_kind = skipCost <= contextPtr->mdContext->mdEpPipeLcu[cu->leafIndex].mergeCost ? SVT_AMD_EP_INTER_SKIP : SVT_AMD_EP_INTER_MERGE;
            }
        }
        uint32_t lg = 0;
        while ((1u << lg) < st->size)
            lg++;
        const uint32_t cuIndex = (st->originY >> lg) * (1u << st->depth) + (st->originX >> lg);
        u->bottom_left_ok = isBottomLeftAvailable(st->depth, cuIndex), u->top_right_ok = isUpperRightAvailable(st->depth, cuIndex);
        u->leaf_index = (uint8_t)cuItr;
        cuItr += DepthOffset[st->depth];
    }
    w->num_cus = (uint8_t)n;
    /* doRecon (:3083-3087): in non-reference pictures of the limitIntra presets an LCU without intra units is not reconstructed (and a
     * skipped unit of it not even predicted); the record then carries flags and coefficients only (bit 1 of dlf_off) */
    int any_intra = 0;
    for (EB_U32 i = 0; i < n; i++)
        any_intra |= w->cu[i].pred_mode == INTRA_MODE;
    const int do_recon = !contextPtr->mdContext->limitIntra || any_intra || pcs->ParentPcsPtr->isUsedAsReferenceFlag || scs->staticConfig.reconEnabled;
    return do_recon ? 1 : 2;
}

/* what EncDecConfigureLcu has set when EncodePass returns (the harness cannot see it before the call only when the process reuses the
 * context for another LCU: it does not, the context is the caller's for the whole call) */
static void fill_rate_inputs(SvtAmdLcuWork *w, const EncDecContext_t *contextPtr)
{
    w->full_lambda = contextPtr->fullLambda;
    w->luma_cbf_bits[0] = contextPtr->mdRateEstimationPtr->lumaCbfBits[0], w->luma_cbf_bits[1] = contextPtr->mdRateEstimationPtr->lumaCbfBits[1];
    w->luma_cbf_bits[2] = contextPtr->mdRateEstimationPtr->lumaCbfBits[(NUMBER_OF_CBF_CASES >> 1)];
    w->luma_cbf_bits[3] = contextPtr->mdRateEstimationPtr->lumaCbfBits[(NUMBER_OF_CBF_CASES >> 1) + 1];
}

/* units' QPs and what the reference left in the TransformUnit_t records, after the call */
static void fill_after(SvtAmdLcuCu *cus, int n, SvtAmdLcuCuResult *out, const PictureControlSet_t *pcs, const LargestCodingUnit_t *lcuPtr)
{
    for (int i = 0; i < n; i++) {
        SvtAmdLcuCu *u = &cus[i];
        const CodingUnit_t *cu = lcuPtr->codedLeafArrayPtr[u->leaf_index];
        const TransformUnit_t *tu = &cu->transformUnitArray[0];
        u->qp = (uint8_t)cu->qp;
        if (u->pred_mode == INTER_MODE && lcuPtr->chromaEncodeMode == CHROMA_MODE_BEST) {
            /* what the pass decided (:4138-4140): a skipped merge unit leaves with skipFlag set and mergeFlag cleared; a merge unit keeps mergeFlag (and is
             * marked skipped only when nothing was coded, :4348-4352) */
            const int merge = cu->predictionUnitArray->mergeFlag, skip = cu->skipFlag;
            u->inter_kind = skip && !merge ? SVT_AMD_EP_INTER_SKIP : merge ? SVT_AMD_EP_INTER_MERGE : SVT_AMD_EP_INTER_AMVP;
        }
        const EB_S8 qs = (EB_S8)CLIP3((EB_S8)MIN_QP_VALUE, (EB_S8)MAX_CHROMA_MAP_QP_VALUE, (EB_S8)(cu->qp + pcs->cbQpOffset + pcs->sliceCbQpOffset));
        u->chroma_qp = MapChromaQp((EB_U8)qs);
        /* a 64x64 unit is alone in its LCU: entry 0 = transformUnitArray[0] (the flags of the four units OR-ed), entries 1..4 = its four
         * 32x32 transform units (tuItr 1..4) */
        for (int k = 0; k < (u->size == 64 ? 5 : 1); k++) {
            SvtAmdLcuCuResult *o = &out[i + k];
            o->cbf[0] = tu[k].lumaCbf, o->cbf[1] = tu[k].cbCbf, o->cbf[2] = tu[k].crCbf;
            for (int p = 0; p < 3; p++)
                o->only_dc[p] = tu[k].isOnlyDc[p], o->nz[p] = tu[k].nzCoefCount[p];
        }
    }
}

/* 10-bit encodes: the source of the LCU is what EncodePassPackLcu left in contextPtr->inputSample16bitBuffer (64-pitch luma, 32-pitch
 * chroma), the reconstruction lives in the 16-bit picture buffers */
static void encode_pass_16bit(SequenceControlSet_t *scs, PictureControlSet_t *pcs, LargestCodingUnit_t *lcuPtr, EB_U32 tbAddr, EB_U32 lcuOriginX,
                              EB_U32 lcuOriginY, EB_U32 lcuQp, EB_BOOL enableSaoFlag, EncDecContext_t *contextPtr)
{
    EpRecord16 *r = (EpRecord16 *)calloc(1, sizeof(*r));
    int kind = 0;
    if (r && !(kind = fill_work((SvtAmdLcuWork *)&r->work /* same head */, scs, pcs, lcuPtr, lcuOriginX, lcuOriginY, contextPtr))) {
        free(r);
        r = NULL;
    }
    if (r)
        dump_inter_inputs(pcs, 1, r->ref_poc);
    __real_EncodePass(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
    if (!r)
        return;
    fill_rate_inputs((SvtAmdLcuWork *)&r->work, contextPtr);
    const EB_U32 lw = MIN(64u, scs->lumaWidth - lcuOriginX), lh = MIN(64u, scs->lumaHeight - lcuOriginY);
    r->magic = EP_DUMP_MAGIC, r->record_size = (uint32_t)sizeof(*r), r->picture_number = pcs->pictureNumber;
    r->width = scs->lumaWidth, r->height = scs->lumaHeight, r->lcu_index = tbAddr;
    r->dlf_off = scs->staticConfig.disableDlfFlag | (kind == 2 ? 2u : 0u);
    fill_after(r->work.cu, r->work.num_cus, r->result.cu, pcs, lcuPtr);
    const EbPictureBufferDesc_t *in = contextPtr->inputSample16bitBuffer;
    for (EB_U32 y = 0; y < lh; y++)
        memcpy(r->work.src_y + y * 64, (const uint16_t *)in->bufferY + (size_t)y * in->strideY, lw * 2);
    for (EB_U32 y = 0; y < lh / 2; y++) {
        memcpy(r->work.src_cb + y * 32, (const uint16_t *)in->bufferCb + (size_t)y * in->strideCb, lw);
        memcpy(r->work.src_cr + y * 32, (const uint16_t *)in->bufferCr + (size_t)y * in->strideCr, lw);
    }
    const EbPictureBufferDesc_t *q = lcuPtr->quantizedCoeff;
    for (int y = 0; y < 64; y++)
        memcpy(r->result.coeff_y + y * 64, (const int16_t *)q->bufferY + (size_t)y * q->strideY, 128);
    for (int y = 0; y < 32; y++) {
        memcpy(r->result.coeff_cb + y * 32, (const int16_t *)q->bufferCb + (size_t)y * q->strideCb, 64);
        memcpy(r->result.coeff_cr + y * 32, (const int16_t *)q->bufferCr + (size_t)y * q->strideCr, 64);
    }
    const EbPictureBufferDesc_t *rec = pcs->ParentPcsPtr->isUsedAsReferenceFlag
                                           ? ((EbReferenceObject_t *)pcs->ParentPcsPtr->referencePictureWrapperPtr->objectPtr)->referencePicture16bit
                                           : pcs->reconPicture16bitPtr;
    for (EB_U32 y = 0; y < lh; y++)
        memcpy(r->result.rec_y + y * 64, (const uint16_t *)rec->bufferY + (size_t)(rec->originY + lcuOriginY + y) * rec->strideY + rec->originX + lcuOriginX, lw * 2);
    for (EB_U32 y = 0; y < lh / 2; y++) {
        memcpy(r->result.rec_cb + y * 32, (const uint16_t *)rec->bufferCb + (size_t)((rec->originY + lcuOriginY) / 2 + y) * rec->strideCb + (rec->originX + lcuOriginX) / 2, lw);
        memcpy(r->result.rec_cr + y * 32, (const uint16_t *)rec->bufferCr + (size_t)((rec->originY + lcuOriginY) / 2 + y) * rec->strideCr + (rec->originX + lcuOriginX) / 2, lw);
    }
    pthread_mutex_lock(&g_lock);
    fwrite(r, sizeof(*r), 1, g_file);
    fflush(g_file);
    pthread_mutex_unlock(&g_lock);
    free(r);
}

void __wrap_EncodePass(SequenceControlSet_t *scs, PictureControlSet_t *pcs, LargestCodingUnit_t *lcuPtr, EB_U32 tbAddr, EB_U32 lcuOriginX,
                       EB_U32 lcuOriginY, EB_U32 lcuQp, EB_BOOL enableSaoFlag, EncDecContext_t *contextPtr)
{
    if (g_state == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_state == 0) {
            const char *path = getenv("SVT_REF_ENCODEPASS_DUMP");
            g_file = path ? fopen(path, "wb") : NULL;
            g_state = g_file ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
    EpRecord *r = NULL;
    int kind = 0;
    if (g_state > 0 && contextPtr->is16bit && contextPtr->colorFormat == EB_YUV420) {
        encode_pass_16bit(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
        return;
    }
    if (g_state > 0 && !contextPtr->is16bit && contextPtr->colorFormat == EB_YUV420 && (r = (EpRecord *)calloc(1, sizeof(*r))) != NULL) {
        if (!(kind = fill_work(&r->work, scs, pcs, lcuPtr, lcuOriginX, lcuOriginY, contextPtr))) {
            free(r);
            r = NULL;
        }
    }
    if (r)
        dump_inter_inputs(pcs, 0, r->ref_poc);
    const EbPictureBufferDesc_t *in = (const EbPictureBufferDesc_t *)pcs->ParentPcsPtr->enhancedPicturePtr;
    const EB_U32 lw = MIN(64u, scs->lumaWidth - lcuOriginX), lh = MIN(64u, scs->lumaHeight - lcuOriginY);
    if (r) {
        for (EB_U32 y = 0; y < lh; y++)
            memcpy(r->work.src_y + y * 64, in->bufferY + (size_t)(in->originY + lcuOriginY + y) * in->strideY + in->originX + lcuOriginX, lw);
        for (EB_U32 y = 0; y < lh / 2; y++) {
            memcpy(r->work.src_cb + y * 32, in->bufferCb + (size_t)((in->originY + lcuOriginY) / 2 + y) * in->strideCb + (in->originX + lcuOriginX) / 2, lw / 2);
            memcpy(r->work.src_cr + y * 32, in->bufferCr + (size_t)((in->originY + lcuOriginY) / 2 + y) * in->strideCr + (in->originX + lcuOriginX) / 2, lw / 2);
        }
    }
    __real_EncodePass(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
    if (!r)
        return;
    r->magic = EP_DUMP_MAGIC, r->record_size = (uint32_t)sizeof(*r), r->picture_number = pcs->pictureNumber;
    r->width = scs->lumaWidth, r->height = scs->lumaHeight, r->lcu_index = tbAddr;
    r->dlf_off = scs->staticConfig.disableDlfFlag | (kind == 2 ? 2u : 0u);
    fill_after(r->work.cu, r->work.num_cus, r->result.cu, pcs, lcuPtr);
    fill_rate_inputs(&r->work, contextPtr);
    const EbPictureBufferDesc_t *q = lcuPtr->quantizedCoeff;
    for (int y = 0; y < 64; y++)
        memcpy(r->result.coeff_y + y * 64, (const int16_t *)q->bufferY + (size_t)y * q->strideY, 128);
    for (int y = 0; y < 32; y++) {
        memcpy(r->result.coeff_cb + y * 32, (const int16_t *)q->bufferCb + (size_t)y * q->strideCb, 64);
        memcpy(r->result.coeff_cr + y * 32, (const int16_t *)q->bufferCr + (size_t)y * q->strideCr, 64);
    }
    const EbPictureBufferDesc_t *rec = pcs->ParentPcsPtr->isUsedAsReferenceFlag
                                           ? ((EbReferenceObject_t *)pcs->ParentPcsPtr->referencePictureWrapperPtr->objectPtr)->referencePicture
                                           : pcs->reconPicturePtr;
    for (EB_U32 y = 0; y < lh; y++)
        memcpy(r->result.rec_y + y * 64, rec->bufferY + (size_t)(rec->originY + lcuOriginY + y) * rec->strideY + rec->originX + lcuOriginX, lw);
    for (EB_U32 y = 0; y < lh / 2; y++) {
        memcpy(r->result.rec_cb + y * 32, rec->bufferCb + (size_t)((rec->originY + lcuOriginY) / 2 + y) * rec->strideCb + (rec->originX + lcuOriginX) / 2, lw / 2);
        memcpy(r->result.rec_cr + y * 32, rec->bufferCr + (size_t)((rec->originY + lcuOriginY) / 2 + y) * rec->strideCr + (rec->originX + lcuOriginX) / 2, lw / 2);
    }
    pthread_mutex_lock(&g_lock);
    fwrite(r, sizeof(*r), 1, g_file);
    fflush(g_file);
    pthread_mutex_unlock(&g_lock);
    free(r);
}
