/*
 * TEST INFRASTRUCTURE ONLY (oracle/): link-time interposers around the REFERENCE's chroma full loop, the pair
 * FullLoop_R (Codec/EbFullLoop.c:579) + CuFullDistortionFastTuMode_R (:873) that the mode decision calls back to back
 * (EbProductCodingLoop.c:4291-4319, :4518-4547).  Compiled only into oracle/_ref/libsvtref.so with
 * -Wl,--wrap=FullLoop_R -Wl,--wrap=CuFullDistortionFastTuMode_R.
 *
 * With SVT_REF_CHROMALOOP_DUMP=<file>, a sample of the call pairs (every SVT_REF_CHROMALOOP_STRIDE-th, default 37) leaves
 * one binary record each: the Cb / Cr residuals of the candidate CU as FullLoop_R found them, the scalars the pair read
 * (chroma qps, slice type, partial-frequency mode, candidate type / intra mode, the CabacCost_t tables) and everything
 * the pair produced (quantised and reconstructed coefficients, non-zero counts, coefficient bits, distortions, cbfs).
 * Only the configuration the BASELINE presets use is recorded: no RDOQ (PM-core, which leaves chroma alone, is), no spatial-SSE full loop, no
 * CABAC-context update.  tests/golden/make_chromaloop_golden.py builds the fixtures.
 * No reference source here; reference headers are included only to read its structs.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbModeDecisionProcess.h"
#include "EbModeDecision.h"
#include "EbFullLoop.h"
#include "EbCabacContextModel.h"

#include "../include/svt_hevc_amd.h"

void __real_FullLoop_R(LargestCodingUnit_t *lcuPtr, ModeDecisionCandidateBuffer_t *candidateBuffer,
                       ModeDecisionContext_t *contextPtr, const CodedUnitStats_t *cuStatsPtr,
                       EbPictureBufferDesc_t *inputPicturePtr, PictureControlSet_t *pcs, EB_U32 componentMask, EB_U32 cbQp,
                       EB_U32 crQp, EB_U32 *cbCountNonZeroCoeffs, EB_U32 *crCountNonZeroCoeffs);
void __real_CuFullDistortionFastTuMode_R(EbPictureBufferDesc_t *inputPicturePtr, EB_U32 inputCbOriginIndex,
                                         LargestCodingUnit_t *lcuPtr, ModeDecisionCandidateBuffer_t *candidateBuffer,
                                         ModeDecisionContext_t *contextPtr, ModeDecisionCandidate_t *candidatePtr,
                                         const CodedUnitStats_t *cuStatsPtr, EB_U64 cbFullDistortion[DIST_CALC_TOTAL],
                                         EB_U64 crFullDistortion[DIST_CALC_TOTAL],
                                         EB_U32 countNonZeroCoeffs[3][MAX_NUM_OF_TU_PER_CU], EB_U32 componentMask,
                                         EB_U64 *cbCoeffBits, EB_U64 *crCoeffBits);

#define CL_DUMP_MAGIC 0x4d524843U /* "CHRM" */

typedef struct ChromaLoopRecord {
    uint32_t magic, record_size;
    uint64_t picture_number;
    uint32_t size, origin_x, origin_y, cb_qp, cr_qp, slice_type, temporal_layer, pf_mode, cand_type, intra_luma_mode;
    uint32_t cbf_in[2], cbf_out[2];
    uint32_t nz_out[2][5];
    uint64_t bits_in[2], bits_out[2], dist_in[2][2], dist_out[2][2];
    SvtAmdCabacCost cost;
    int16_t residual[2][32 * 32], quant[2][32 * 32], recon[2][32 * 32]; /* (size/2)^2 used, row pitch = size/2 */
} ChromaLoopRecord;

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static FILE *g_file;
static int g_state, g_stride = 37;
static unsigned long g_calls;
static __thread ChromaLoopRecord *t_pending; /* record opened by FullLoop_R, closed by CuFullDistortionFastTuMode_R */
static __thread const ModeDecisionCandidateBuffer_t *t_pending_buffer;

static void grab(int16_t *dst, const void *plane, uint32_t origin, uint32_t csize)
{
    const int16_t *src = (const int16_t *)plane + origin;
    for (uint32_t y = 0; y < csize; y++)
        memcpy(dst + y * csize, src + (size_t)y * 32, csize * sizeof(int16_t));
}

void __wrap_FullLoop_R(LargestCodingUnit_t *lcuPtr, ModeDecisionCandidateBuffer_t *candidateBuffer,
                       ModeDecisionContext_t *contextPtr, const CodedUnitStats_t *cuStatsPtr,
                       EbPictureBufferDesc_t *inputPicturePtr, PictureControlSet_t *pcs, EB_U32 componentMask, EB_U32 cbQp,
                       EB_U32 crQp, EB_U32 *cbCountNonZeroCoeffs, EB_U32 *crCountNonZeroCoeffs)
{
    if (g_state == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_state == 0) {
            const char *path = getenv("SVT_REF_CHROMALOOP_DUMP"), *st = getenv("SVT_REF_CHROMALOOP_STRIDE");
            g_file = path ? fopen(path, "wb") : NULL;
            if (st && atoi(st) > 0)
                g_stride = atoi(st);
            g_state = g_file ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
    int take = 0;
    if (g_state > 0 && (!contextPtr->rdoqPmCoreMethod || contextPtr->rdoqPmCoreMethod == EB_PMCORE) && !contextPtr->spatialSseFullLoop &&
        !contextPtr->coeffCabacUpdate &&
        componentMask == PICTURE_BUFFER_DESC_CHROMA_MASK && candidateBuffer->residualQuantCoeffPtr->strideCb == 32) {
        pthread_mutex_lock(&g_lock);
        take = (g_calls++ % (unsigned long)g_stride) == 0;
        pthread_mutex_unlock(&g_lock);
    }
    free(t_pending);
    t_pending = NULL;
    if (!take) {
        __real_FullLoop_R(lcuPtr, candidateBuffer, contextPtr, cuStatsPtr, inputPicturePtr, pcs, componentMask, cbQp, crQp,
                          cbCountNonZeroCoeffs, crCountNonZeroCoeffs);
        return;
    }
    ChromaLoopRecord *r = (ChromaLoopRecord *)calloc(1, sizeof(*r));
    const uint32_t size = cuStatsPtr->size, cs = size >> 1;
    const uint32_t origin = size == 64 ? 0 : (cuStatsPtr->originX + cuStatsPtr->originY * 32) >> 1;
    const ModeDecisionCandidate_t *c = candidateBuffer->candidatePtr;
    r->magic = CL_DUMP_MAGIC, r->record_size = (uint32_t)sizeof(*r);
    r->picture_number = pcs->pictureNumber;
    r->size = size, r->origin_x = cuStatsPtr->originX, r->origin_y = cuStatsPtr->originY, r->cb_qp = cbQp, r->cr_qp = crQp;
    r->slice_type = pcs->sliceType, r->temporal_layer = pcs->temporalLayerIndex;
    r->pf_mode = contextPtr->pfMdMode, r->cand_type = c->type, r->intra_luma_mode = c->intraLumaMode;
    memcpy(&r->cost, contextPtr->CabacCost, sizeof(r->cost));
    grab(r->residual[0], candidateBuffer->residualQuantCoeffPtr->bufferCb, origin, cs);
    grab(r->residual[1], candidateBuffer->residualQuantCoeffPtr->bufferCr, origin, cs);

    __real_FullLoop_R(lcuPtr, candidateBuffer, contextPtr, cuStatsPtr, inputPicturePtr, pcs, componentMask, cbQp, crQp,
                      cbCountNonZeroCoeffs, crCountNonZeroCoeffs);

    grab(r->quant[0], candidateBuffer->residualQuantCoeffPtr->bufferCb, origin, cs);
    grab(r->quant[1], candidateBuffer->residualQuantCoeffPtr->bufferCr, origin, cs);
    grab(r->recon[0], candidateBuffer->reconCoeffPtr->bufferCb, origin, cs);
    grab(r->recon[1], candidateBuffer->reconCoeffPtr->bufferCr, origin, cs);
    t_pending = r;
    t_pending_buffer = candidateBuffer;
}

void __wrap_CuFullDistortionFastTuMode_R(EbPictureBufferDesc_t *inputPicturePtr, EB_U32 inputCbOriginIndex,
                                         LargestCodingUnit_t *lcuPtr, ModeDecisionCandidateBuffer_t *candidateBuffer,
                                         ModeDecisionContext_t *contextPtr, ModeDecisionCandidate_t *candidatePtr,
                                         const CodedUnitStats_t *cuStatsPtr, EB_U64 cbFullDistortion[DIST_CALC_TOTAL],
                                         EB_U64 crFullDistortion[DIST_CALC_TOTAL],
                                         EB_U32 countNonZeroCoeffs[3][MAX_NUM_OF_TU_PER_CU], EB_U32 componentMask,
                                         EB_U64 *cbCoeffBits, EB_U64 *crCoeffBits)
{
    ChromaLoopRecord *r = t_pending;
    if (r && (t_pending_buffer != candidateBuffer || componentMask != PICTURE_BUFFER_DESC_CHROMA_MASK)) {
        free(r);
        r = NULL;
    }
    t_pending = NULL;
    if (r) {
        r->cbf_in[0] = candidatePtr->cbCbf, r->cbf_in[1] = candidatePtr->crCbf;
        r->bits_in[0] = *cbCoeffBits, r->bits_in[1] = *crCoeffBits;
        r->dist_in[0][0] = cbFullDistortion[0], r->dist_in[0][1] = cbFullDistortion[1];
        r->dist_in[1][0] = crFullDistortion[0], r->dist_in[1][1] = crFullDistortion[1];
    }
    __real_CuFullDistortionFastTuMode_R(inputPicturePtr, inputCbOriginIndex, lcuPtr, candidateBuffer, contextPtr, candidatePtr,
                                        cuStatsPtr, cbFullDistortion, crFullDistortion, countNonZeroCoeffs, componentMask,
                                        cbCoeffBits, crCoeffBits);
    if (!r)
        return;
    r->cbf_out[0] = candidatePtr->cbCbf, r->cbf_out[1] = candidatePtr->crCbf;
    r->bits_out[0] = *cbCoeffBits, r->bits_out[1] = *crCoeffBits;
    r->dist_out[0][0] = cbFullDistortion[0], r->dist_out[0][1] = cbFullDistortion[1];
    r->dist_out[1][0] = crFullDistortion[0], r->dist_out[1][1] = crFullDistortion[1];
    memcpy(r->nz_out[0], countNonZeroCoeffs[1], sizeof(r->nz_out[0]));
    memcpy(r->nz_out[1], countNonZeroCoeffs[2], sizeof(r->nz_out[1]));
    pthread_mutex_lock(&g_lock);
    fwrite(r, sizeof(*r), 1, g_file);
    fflush(g_file);
    pthread_mutex_unlock(&g_lock);
    free(r);
}
