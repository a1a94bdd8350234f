/*
 * integration/svt_hook_encdec.c - the reference-side binding of the device-resident encode pass (SURVEY 8b `hip_encdec_segment`,
 * include/svt_hevc_amd.h "Device-resident encode pass"), enabled with SVT_HOOK_ENCODEPASS=1.
 *
 * EncodePass (Codec/EbCodingLoop.c:2989) is interposed with -Wl,--wrap=EncodePass.  For an LCU inside what this revision of
 * svt_amd_encode_lcus() / svt_amd_encode_lcus16() covers - 4:2:0, 8-bit or 10-bit (EncodePass with is16bit), every coding unit an
 * intra 2Nx2N unit of 8..32, no delta-QP / masking tools, plain quantiser - the binding
 *   1. converts the LCU's final coding-unit tree (LargestCodingUnit_t.codedLeafArrayPtr) and its source samples into the input
 *      contract SvtAmdLcuWork,
 *   2. makes ONE device call: intra reference + prediction, residual, transform, quantiser, inverse transform and reconstruction of
 *      all units and planes of the LCU run on the MI355X against the picture's un-deblocked reconstruction, which stays in HBM,
 *   3. lets the reference's EncodePass run for its bookkeeping (cbf / DC flags, neighbour arrays, boundary strengths, deblocking,
 *      SAO, coefficient buffer for entropy coding) with every compute leaf it reaches answered from the output contract
 *      SvtAmdLcuResult: the intra generator / predictor table slots and PictureResidual / EstimateTransform return at once,
 *      UnifiedQuantizeInvQuantize hands back the device's coefficients, the EncodeGenerateRecon table slot the device's samples.
 * LCUs outside that (inter units, intra 4x4, PM-core quantiser at encMode <= 4, ...) are encoded by the reference code; their last
 * row / column and edge mode types (the ep* neighbour arrays after the call) are handed to the device picture before the next
 * device-encoded LCU of the picture needs them.  No fallback on errors: any failure of the HIP library aborts the encoder.
 *
 * Contains no reference source.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbEncDecProcess.h"
#include "EbModeDecisionProcess.h"
#include "EbReferenceObject.h"
#include "EbCodingUnit.h"
#include "EbTransformUnit.h"
#include "EbTransforms.h"
#include "EbNeighborArrays.h"
#include "EbUtility.h"
#include "EbAvailability.h"
#include "EbPictureOperators.h"

#include "svt_hook_internal.h"

#define EP_LANES 8     /* device contexts (stream + staging) EncDec threads share */
#define EP_PICTURES 64 /* pictures in flight (PictureControlSet_t objects of the EncDec pool) */

void __real_EncodePass(SequenceControlSet_t *scs, PictureControlSet_t *pcs, LargestCodingUnit_t *lcuPtr, EB_U32 tbAddr, EB_U32 lcuOriginX,
                       EB_U32 lcuOriginY, EB_U32 lcuQp, EB_BOOL enableSaoFlag, EncDecContext_t *contextPtr);
void __real_PictureResidual(EB_U8 *input, EB_U32 inputStride, EB_U8 *pred, EB_U32 predStride, EB_S16 *residual, EB_U32 residualStride,
                            EB_U32 areaWidth, EB_U32 areaHeight);
EB_ERRORTYPE __real_EstimateTransform(EB_S16 *residualBuffer, EB_U32 residualStride, EB_S16 *coeffBuffer, EB_U32 coeffStride, EB_U32 transformSize,
                                      EB_S16 *transformInnerArrayPtr, EB_U32 bitIncrement, EB_BOOL dstTansformFlag,
                                      EB_TRANS_COEFF_SHAPE transCoeffShape);

typedef struct {
    const PictureControlSet_t *pcs;
    uint64_t picture_plus1;      /* picture the device picture was begun for */
    SvtAmdEncDecPicture *pic;
    pthread_mutex_t lock;        /* pending list + the device put of it */
    void *pending;               /* host-encoded LCUs not handed to the device yet: SvtAmdLcuBorder[] or SvtAmdLcuBorder16[] */
    int npending, cap, wide;
} EpPictureEntry;

typedef struct {
    union { /* the heads of the 8- and the 16-bit contract are the same */
        SvtAmdLcuWork work;
        SvtAmdLcuWork16 work16;
    };
    union {
        SvtAmdLcuResult res;
        SvtAmdLcuResult16 res16;
    };
    const LargestCodingUnit_t *lcu;
    int wide; /* 10-bit encode: the 16-bit members are live */
} EpServe;
_Static_assert(offsetof(SvtAmdLcuWork, src_y) == offsetof(SvtAmdLcuWork16, src_y) && offsetof(SvtAmdLcuResult, rec_y) == offsetof(SvtAmdLcuResult16, rec_y),
               "contract heads");
void EncodePassPackLcu(SequenceControlSet_t *sequenceControlSetPtr, EbPictureBufferDesc_t *inputPicture, EncDecContext_t *contextPtr,
                       EB_U32 lcuOriginX, EB_U32 lcuOriginY, EB_U32 lcuWidth, EB_U32 lcuHeight); /* EbCodingLoop.c:2867 */

__thread int svt_hook_ep_active;
static __thread EpServe *t_serve;

static pthread_mutex_t g_ep_lock = PTHREAD_MUTEX_INITIALIZER; /* picture table + lane pool */
static pthread_cond_t g_ep_cv = PTHREAD_COND_INITIALIZER;
static EpPictureEntry g_ep_pic[EP_PICTURES];
static SvtAmdContext *g_ep_lane[EP_LANES];
static int g_ep_lane_busy[EP_LANES];
static int g_ep_state; /* 0 unknown, 1 on, -1 off */
static unsigned long g_ep_gpu, g_ep_cpu_units, g_ep_cpu_tools, g_ep_cpu_format, g_ep_borders, g_ep_puts;

static SvtAmdContext *lane_claim(SvtAmdContext *root)
{
    pthread_mutex_lock(&g_ep_lock);
    for (;;) {
        for (int i = 0; i < EP_LANES; i++)
            if (!g_ep_lane_busy[i]) {
                if (!g_ep_lane[i] && svt_amd_context_fork(root, &g_ep_lane[i]))
                    svt_hook_die("svt_amd_context_fork (encode pass)");
                g_ep_lane_busy[i] = 1;
                pthread_mutex_unlock(&g_ep_lock);
                return g_ep_lane[i];
            }
        pthread_cond_wait(&g_ep_cv, &g_ep_lock);
    }
}

static void lane_release(SvtAmdContext *lane)
{
    pthread_mutex_lock(&g_ep_lock);
    for (int i = 0; i < EP_LANES; i++)
        if (g_ep_lane[i] == lane)
            g_ep_lane_busy[i] = 0;
    pthread_cond_signal(&g_ep_cv);
    pthread_mutex_unlock(&g_ep_lock);
}

/* the device picture of this PictureControlSet_t, begun for its current picture */
static EpPictureEntry *picture_entry(SvtAmdContext *lane, const SequenceControlSet_t *scs, const PictureControlSet_t *pcs, int wide)
{
    EpPictureEntry *e = NULL;
    pthread_mutex_lock(&g_ep_lock);
    for (int i = 0; i < EP_PICTURES && !e; i++)
        if (g_ep_pic[i].pcs == pcs)
            e = &g_ep_pic[i];
    for (int i = 0; i < EP_PICTURES && !e; i++)
        if (!g_ep_pic[i].pcs) {
            e = &g_ep_pic[i];
            e->pcs = pcs;
            pthread_mutex_init(&e->lock, NULL);
            e->wide = wide;
            if (svt_amd_encdec_picture_create(lane, (uint16_t)scs->lumaWidth, (uint16_t)scs->lumaHeight, wide ? 2 : 1, &e->pic))
                svt_hook_die("svt_amd_encdec_picture_create");
            e->cap = (int)(((scs->lumaWidth + 63u) / 64u) * ((scs->lumaHeight + 63u) / 64u));
            e->pending = malloc((wide ? sizeof(SvtAmdLcuBorder16) : sizeof(SvtAmdLcuBorder)) * (size_t)e->cap);
            if (!e->pending)
                svt_hook_die("out of memory (encode-pass border list)");
        }
    if (!e)
        svt_hook_die("encode pass: more PictureControlSet_t objects than EP_PICTURES");
    if (e->picture_plus1 != pcs->pictureNumber + 1) { /* first LCU of a new picture in this object: nothing coded yet */
        if (svt_amd_encdec_picture_begin(lane, e->pic))
            svt_hook_die("svt_amd_encdec_picture_begin");
        e->picture_plus1 = pcs->pictureNumber + 1;
        e->npending = 0;
    }
    pthread_mutex_unlock(&g_ep_lock);
    return e;
}

/* EncDec input contract: the coded leaves of the LCU in Z order.  Returns 0 when a unit is outside what the device call covers. */
static int fill_work(SvtAmdLcuWork *w, const SequenceControlSet_t *scs, const PictureControlSet_t *pcs, const LargestCodingUnit_t *lcuPtr,
                     EB_U32 lcuOriginX, EB_U32 lcuOriginY)
{
    memset(w, 0, offsetof(SvtAmdLcuWork, src_y));
    w->lcu_x = (uint16_t)lcuOriginX, w->lcu_y = (uint16_t)lcuOriginY;
    w->slice_type = (uint8_t)pcs->sliceType, w->temporal_layer = pcs->temporalLayerIndex;
    w->constrained_intra = pcs->constrainedIntraFlag, w->strong_smoothing = scs->enableStrongIntraSmoothing;
    w->tile_left = lcuPtr->lcuEdgeInfoPtr->tileLeftEdgeFlag, w->tile_top = lcuPtr->lcuEdgeInfoPtr->tileTopEdgeFlag;
    w->tile_right = lcuPtr->lcuEdgeInfoPtr->tileRightEdgeFlag;
    /* without the delta-QP tools every unit is coded at the picture's QP (EbCodingLoop.c:3214, :3238-3246) */
    const EB_U8 qp = pcs->pictureQp;
    const EB_U8 chromaQp = MapChromaQp((EB_U8)CLIP3((EB_S8)MIN_QP_VALUE, (EB_S8)MAX_CHROMA_MAP_QP_VALUE, (EB_S8)(qp + pcs->cbQpOffset + pcs->sliceCbQpOffset)));
    EB_U32 cuItr = 0, n = 0;
    while (cuItr < CU_MAX_COUNT) {
        const CodingUnit_t *cu = lcuPtr->codedLeafArrayPtr[cuItr];
        if (cu->splitFlag) {
            cuItr++;
            continue;
        }
        const CodedUnitStats_t *st = GetCodedUnitStats(cuItr);
        if (cu->predictionModeFlag != INTRA_MODE || cu->predictionUnitArray->intraLumaMode == EB_INTRA_MODE_4x4 || st->size > 32 || st->size < 8 ||
            n >= SVT_AMD_LCU_MAX_CUS)
            return 0;
        SvtAmdLcuCu *u = &w->cu[n++];
        u->x = st->originX, u->y = st->originY, u->size = st->size, u->pred_mode = (uint8_t)cu->predictionModeFlag;
        u->intra_luma_mode = (uint8_t)cu->predictionUnitArray->intraLumaMode;
        const uint32_t lg = (uint32_t)Log2f(st->size);
        const uint32_t cuIndex = (st->originY >> lg) * (1u << st->depth) + (st->originX >> lg);
        u->bottom_left_ok = isBottomLeftAvailable(st->depth, cuIndex), u->top_right_ok = isUpperRightAvailable(st->depth, cuIndex);
        u->qp = qp, u->chroma_qp = chromaQp, u->leaf_index = (uint8_t)cuItr, u->dz_offset = 0;
        cuItr += DepthOffset[st->depth];
    }
    w->num_cus = (uint8_t)n;
    return n > 0;
}

/* what a host-encoded LCU leaves for its neighbours: the top / left entries of the ep* neighbour arrays over its extent */
static void border_from_neighbour_arrays(void *out, int wide, const PictureControlSet_t *pcs, EB_U32 tileIdx, EB_U32 x0, EB_U32 y0, EB_U32 lw,
                                         EB_U32 lh)
{
    NeighborArrayUnit_t *mode = pcs->epModeTypeNeighborArray[tileIdx];
    NeighborArrayUnit_t *na[3] = {wide ? pcs->epLumaReconNeighborArray16bit[tileIdx] : pcs->epLumaReconNeighborArray[tileIdx],
                                  wide ? pcs->epCbReconNeighborArray16bit[tileIdx] : pcs->epCbReconNeighborArray[tileIdx],
                                  wide ? pcs->epCrReconNeighborArray16bit[tileIdx] : pcs->epCrReconNeighborArray[tileIdx]};
    uint8_t *mb, *mr;
    void *dst[6]; /* bottom / right of Y, Cb, Cr */
    if (wide) {
        SvtAmdLcuBorder16 *b = (SvtAmdLcuBorder16 *)out;
        memset(b, 0, sizeof(*b));
        b->lcu_x = (uint16_t)x0, b->lcu_y = (uint16_t)y0, mb = b->mode_bottom, mr = b->mode_right;
        dst[0] = b->bottom_y, dst[1] = b->right_y, dst[2] = b->bottom_cb, dst[3] = b->right_cb, dst[4] = b->bottom_cr, dst[5] = b->right_cr;
    } else {
        SvtAmdLcuBorder *b = (SvtAmdLcuBorder *)out;
        memset(b, 0, sizeof(*b));
        b->lcu_x = (uint16_t)x0, b->lcu_y = (uint16_t)y0, mb = b->mode_bottom, mr = b->mode_right;
        dst[0] = b->bottom_y, dst[1] = b->right_y, dst[2] = b->bottom_cb, dst[3] = b->right_cb, dst[4] = b->bottom_cr, dst[5] = b->right_cr;
    }
    for (EB_U32 k = 0; k < lw / 4; k++)
        mb[k] = mode->topArray[GetNeighborArrayUnitTopIndex(mode, x0 + 4 * k)];
    for (EB_U32 k = 0; k < lh / 4; k++)
        mr[k] = mode->leftArray[GetNeighborArrayUnitLeftIndex(mode, y0 + 4 * k)];
    const size_t bps = wide ? 2 : 1;
    for (int p = 0; p < 3; p++) {
        const EB_U32 sh = p ? 1 : 0;
        memcpy(dst[2 * p], na[p]->topArray + (size_t)(x0 >> sh) * bps, (size_t)(lw >> sh) * bps);
        memcpy(dst[2 * p + 1], na[p]->leftArray + (size_t)(y0 >> sh) * bps, (size_t)(lh >> sh) * bps);
    }
}

void __wrap_EncodePass(SequenceControlSet_t *scs, PictureControlSet_t *pcs, LargestCodingUnit_t *lcuPtr, EB_U32 tbAddr, EB_U32 lcuOriginX,
                       EB_U32 lcuOriginY, EB_U32 lcuQp, EB_BOOL enableSaoFlag, EncDecContext_t *contextPtr)
{
    if (g_ep_state == 0)
        g_ep_state = getenv("SVT_HOOK_ENCODEPASS") ? 1 : -1;
    if (g_ep_state < 0 || contextPtr->colorFormat != EB_YUV420 || (scs->lumaWidth & 7) || (scs->lumaHeight & 7)) {
        if (g_ep_state > 0)
            __atomic_add_fetch(&g_ep_cpu_format, 1, __ATOMIC_RELAXED);
        __real_EncodePass(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
        return;
    }
    SvtAmdContext *root = svt_hook_device((uint16_t)scs->lumaWidth, (uint16_t)scs->lumaHeight);
    SvtAmdContext *lane = lane_claim(root);
    const int wide = contextPtr->is16bit != 0; /* 10-bit encode: 16-bit samples, EncodeLoop16bit */
    EpPictureEntry *e = picture_entry(lane, scs, pcs, wide);
    const EB_U32 lw = MIN(64u, scs->lumaWidth - lcuOriginX), lh = MIN(64u, scs->lumaHeight - lcuOriginY);
    if (!t_serve && !(t_serve = (EpServe *)malloc(sizeof(EpServe))))
        svt_hook_die("out of memory (encode-pass staging)");
    /* tools that change a unit's QP, dead zone, coefficient shape or quantiser are outside this revision */
    const int tools = scs->staticConfig.improveSharpness || scs->staticConfig.bitRateReduction || scs->staticConfig.segmentOvEnabled ||
                      contextPtr->mdContext->rdoqPmCoreMethod != EB_NO_RDOQ;
    const int units = !tools && fill_work(&t_serve->work, scs, pcs, lcuPtr, lcuOriginX, lcuOriginY);
    if (!units) {
        lane_release(lane);
        __atomic_add_fetch(tools ? &g_ep_cpu_tools : &g_ep_cpu_units, 1, __ATOMIC_RELAXED);
        __real_EncodePass(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
        pthread_mutex_lock(&e->lock);
        if (e->npending >= e->cap)
            svt_hook_die("encode pass: border list overflow");
        border_from_neighbour_arrays((uint8_t *)e->pending + (size_t)e->npending++ * (wide ? sizeof(SvtAmdLcuBorder16) : sizeof(SvtAmdLcuBorder)), wide,
                                     pcs, contextPtr->encDecTileIndex, lcuOriginX, lcuOriginY, lw, lh);
        __atomic_add_fetch(&g_ep_borders, 1, __ATOMIC_RELAXED);
        pthread_mutex_unlock(&e->lock);
        return;
    }
    /* source samples of the LCU */
    EbPictureBufferDesc_t *in = (EbPictureBufferDesc_t *)pcs->ParentPcsPtr->enhancedPicturePtr;
    SvtAmdLcuWork *w = &t_serve->work;
    t_serve->wide = wide;
    if (wide) { /* the reference packs the 8 + 2 bit planes of the LCU into its 16-bit LCU buffer at the top of EncodePass; do it now */
        EncodePassPackLcu(scs, in, contextPtr, lcuOriginX, lcuOriginY, lw, lh);
        const EbPictureBufferDesc_t *s16 = contextPtr->inputSample16bitBuffer;
        SvtAmdLcuWork16 *w16 = &t_serve->work16;
        for (EB_U32 y = 0; y < lh; y++)
            memcpy(w16->src_y + y * 64, (const uint16_t *)s16->bufferY + (size_t)y * s16->strideY, lw * 2);
        for (EB_U32 y = 0; y < lh / 2; y++) {
            memcpy(w16->src_cb + y * 32, (const uint16_t *)s16->bufferCb + (size_t)y * s16->strideCb, lw);
            memcpy(w16->src_cr + y * 32, (const uint16_t *)s16->bufferCr + (size_t)y * s16->strideCr, lw);
        }
    } else {
        for (EB_U32 y = 0; y < lh; y++)
            memcpy(w->src_y + y * 64, in->bufferY + (size_t)(in->originY + lcuOriginY + y) * in->strideY + in->originX + lcuOriginX, lw);
        for (EB_U32 y = 0; y < lh / 2; y++) {
            memcpy(w->src_cb + y * 32, in->bufferCb + (size_t)((in->originY + lcuOriginY) / 2 + y) * in->strideCb + (in->originX + lcuOriginX) / 2, lw / 2);
            memcpy(w->src_cr + y * 32, in->bufferCr + (size_t)((in->originY + lcuOriginY) / 2 + y) * in->strideCr + (in->originX + lcuOriginX) / 2, lw / 2);
        }
    }
    /* LCUs the host encoded since the last device call enter the device picture first (under the picture's lock, so that a second
     * thread's LCU cannot overtake a put it depends on) */
    pthread_mutex_lock(&e->lock);
    if (e->npending) {
        if (wide ? svt_amd_encdec_picture_put_borders16(lane, e->pic, (const SvtAmdLcuBorder16 *)e->pending, e->npending)
                 : svt_amd_encdec_picture_put_borders(lane, e->pic, (const SvtAmdLcuBorder *)e->pending, e->npending))
            svt_hook_die("svt_amd_encdec_picture_put_borders");
        e->npending = 0;
        __atomic_add_fetch(&g_ep_puts, 1, __ATOMIC_RELAXED);
    }
    pthread_mutex_unlock(&e->lock);
    if (wide ? svt_amd_encode_lcus16(lane, e->pic, &t_serve->work16, 1, &t_serve->res16) : svt_amd_encode_lcus(lane, e->pic, w, 1, &t_serve->res))
        svt_hook_die("svt_amd_encode_lcus");
    lane_release(lane);
    __atomic_add_fetch(&g_ep_gpu, 1, __ATOMIC_RELAXED);
    t_serve->lcu = lcuPtr;
    svt_hook_ep_active = 1;
    __real_EncodePass(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
    svt_hook_ep_active = 0;
}

void __wrap_PictureResidual(EB_U8 *input, EB_U32 inputStride, EB_U8 *pred, EB_U32 predStride, EB_S16 *residual, EB_U32 residualStride,
                            EB_U32 areaWidth, EB_U32 areaHeight)
{
    if (!svt_hook_ep_active)
        __real_PictureResidual(input, inputStride, pred, predStride, residual, residualStride, areaWidth, areaHeight);
}

void __real_PictureResidual16bit(EB_U16 *input, EB_U32 inputStride, EB_U16 *pred, EB_U32 predStride, EB_S16 *residual, EB_U32 residualStride,
                                 EB_U32 areaWidth, EB_U32 areaHeight);
void __wrap_PictureResidual16bit(EB_U16 *input, EB_U32 inputStride, EB_U16 *pred, EB_U32 predStride, EB_S16 *residual, EB_U32 residualStride,
                                 EB_U32 areaWidth, EB_U32 areaHeight)
{
    if (!svt_hook_ep_active)
        __real_PictureResidual16bit(input, inputStride, pred, predStride, residual, residualStride, areaWidth, areaHeight);
}

/* EncodeLoop16bit's transform (EbCodingLoop.c:1321; the C table of EncodeTransform holds the same Estimate forms for 10-bit) */
EB_ERRORTYPE __real_EncodeTransform(EB_S16 *residualBuffer, EB_U32 residualStride, EB_S16 *coeffBuffer, EB_U32 coeffStride, EB_U32 transformSize,
                                    EB_S16 *transformInnerArrayPtr, EB_U32 bitIncrement, EB_BOOL dstTransformFlag, EB_TRANS_COEFF_SHAPE transCoeffShape);
EB_ERRORTYPE __wrap_EncodeTransform(EB_S16 *residualBuffer, EB_U32 residualStride, EB_S16 *coeffBuffer, EB_U32 coeffStride, EB_U32 transformSize,
                                    EB_S16 *transformInnerArrayPtr, EB_U32 bitIncrement, EB_BOOL dstTransformFlag, EB_TRANS_COEFF_SHAPE transCoeffShape)
{
    if (svt_hook_ep_active)
        return EB_ErrorNone;
    return __real_EncodeTransform(residualBuffer, residualStride, coeffBuffer, coeffStride, transformSize, transformInnerArrayPtr, bitIncrement,
                                  dstTransformFlag, transCoeffShape);
}

EB_ERRORTYPE __wrap_EstimateTransform(EB_S16 *residualBuffer, EB_U32 residualStride, EB_S16 *coeffBuffer, EB_U32 coeffStride, EB_U32 transformSize,
                                      EB_S16 *transformInnerArrayPtr, EB_U32 bitIncrement, EB_BOOL dstTansformFlag,
                                      EB_TRANS_COEFF_SHAPE transCoeffShape)
{
    if (svt_hook_ep_active)
        return EB_ErrorNone;
    return __real_EstimateTransform(residualBuffer, residualStride, coeffBuffer, coeffStride, transformSize, transformInnerArrayPtr, bitIncrement,
                                    dstTansformFlag, transCoeffShape);
}

/* the unit of the served LCU the reference is at */
static int serve_unit(const EncDecContext_t *contextPtr)
{
    const SvtAmdLcuWork *w = &t_serve->work;
    const EB_U32 x = contextPtr->cuStats->originX, y = contextPtr->cuStats->originY;
    for (int i = 0; i < w->num_cus; i++)
        if (w->cu[i].x == x && w->cu[i].y == y && w->cu[i].size == contextPtr->cuStats->size)
            return i;
    svt_hook_die("encode pass: the reference reached a unit the device call did not hold");
    return -1;
}

void svt_hook_ep_quantize(EncDecContext_t *contextPtr, EB_S16 *quantCoeff, EB_S16 *reconCoeff, EB_U32 coeffStride, EB_U32 qp, EB_U32 areaSize,
                          EB_U32 *nz, EB_U32 shape, EB_U32 cleanSparse, EB_U32 masking, EB_U32 enableCbflag, EB_U32 contouring, EB_U32 dZoffset)
{
    const int i = serve_unit(contextPtr);
    const SvtAmdLcuCu *u = &t_serve->work.cu[i];
    const EbPictureBufferDesc_t *q = t_serve->lcu->quantizedCoeff;
    /* which plane: the destination lies in one of the LCU's three coefficient planes, at the unit's position */
    int p = -1;
    const EB_S16 *base[3] = {(const EB_S16 *)q->bufferY, (const EB_S16 *)q->bufferCb, (const EB_S16 *)q->bufferCr};
    for (int k = 0; k < 3; k++) {
        const EB_U32 pitch = k ? 32 : 64, off = k ? (u->y / 2u) * pitch + u->x / 2u : u->y * pitch + u->x;
        if (quantCoeff == base[k] + off)
            p = k;
    }
    const EB_U32 n = p > 0 ? u->size / 2u : u->size;
    if (p < 0 || areaSize != n || coeffStride != (p ? 32u : 64u) || qp != (EB_U32)(p ? u->chroma_qp : u->qp) + (t_serve->wide ? 12u : 0u) || shape ||
        cleanSparse || masking ||
        enableCbflag || contouring || dZoffset || !nz)
        svt_hook_die("encode pass: the reference's quantiser call differs from what the device encoded (unit, plane, QP or tool flags)");
    const int16_t *src = (p == 0 ? t_serve->res.coeff_y : p == 1 ? t_serve->res.coeff_cb : t_serve->res.coeff_cr) + (quantCoeff - base[p]);
    for (EB_U32 r = 0; r < n; r++)
        memcpy(quantCoeff + r * coeffStride, src + r * coeffStride, n * sizeof(int16_t));
    *nz = t_serve->res.cu[i].nz[p];
    reconCoeff[0] = t_serve->res.cu[i].only_dc[p]; /* the caller's isOnlyDc test (EbCodingLoop.c:792, 879, 1000) reads the DC term */
}

void svt_hook_ep_recon(EncDecContext_t *contextPtr, EB_U32 originX, EB_U32 originY, EB_U32 tuSize, EbPictureBufferDesc_t *recon)
{
    const int i = serve_unit(contextPtr);
    const SvtAmdLcuCu *u = &t_serve->work.cu[i];
    if (tuSize != u->size || (originX & 63) != u->x || (originY & 63) != u->y)
        svt_hook_die("encode pass: reconstruction call for another unit");
    const size_t bps = t_serve->wide ? 2 : 1;
    const uint8_t *ry = (const uint8_t *)t_serve->res.rec_y, *rcb, *rcr; /* rec_y sits at the same offset in both contracts */
    if (t_serve->wide)
        rcb = (const uint8_t *)t_serve->res16.rec_cb, rcr = (const uint8_t *)t_serve->res16.rec_cr;
    else
        rcb = t_serve->res.rec_cb, rcr = t_serve->res.rec_cr;
    EB_U8 *y = recon->bufferY + ((size_t)(recon->originY + originY) * recon->strideY + recon->originX + originX) * bps;
    for (EB_U32 r = 0; r < tuSize; r++)
        memcpy(y + (size_t)r * recon->strideY * bps, ry + ((size_t)(u->y + r) * 64 + u->x) * bps, tuSize * bps);
    EB_U8 *cb = recon->bufferCb + ((size_t)((recon->originY + originY) / 2) * recon->strideCb + (recon->originX + originX) / 2) * bps;
    EB_U8 *cr = recon->bufferCr + ((size_t)((recon->originY + originY) / 2) * recon->strideCr + (recon->originX + originX) / 2) * bps;
    for (EB_U32 r = 0; r < tuSize / 2; r++) {
        memcpy(cb + (size_t)r * recon->strideCb * bps, rcb + ((size_t)(u->y / 2 + r) * 32 + u->x / 2) * bps, tuSize / 2 * bps);
        memcpy(cr + (size_t)r * recon->strideCr * bps, rcr + ((size_t)(u->y / 2 + r) * 32 + u->x / 2) * bps, tuSize / 2 * bps);
    }
}

void svt_hook_encdec_report(FILE *out)
{
    if (g_ep_state <= 0)
        return;
    fprintf(out, "svt_hook_me: encode pass: %lu LCUs encoded on the GPU (one call each); left to the reference code: %lu LCUs with units outside "
                 "the device call, %lu under tools outside it, %lu in another sample format; %lu host LCU borders handed over in %lu calls\n",
            g_ep_gpu, g_ep_cpu_units, g_ep_cpu_tools, g_ep_cpu_format, g_ep_borders, g_ep_puts);
}
