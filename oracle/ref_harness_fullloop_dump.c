/*
 * TEST INFRASTRUCTURE ONLY (oracle/): link-time interposer around the REFERENCE's ProductFullLoop
 * (Codec/EbFullLoop.c:185), compiled only into oracle/_ref/libsvtref.so with -Wl,--wrap=ProductFullLoop.
 *
 * With SVT_REF_FULLLOOP_DUMP=<file>, a sample of the calls (every SVT_REF_FULLLOOP_STRIDE-th, default 53) leaves one
 * binary record each: the luma residual of the candidate CU as the call found it, the scalars the call read (qp,
 * slice type, partial-frequency mode, lambda, candidate type / intra mode, the two luma-cbf bit costs, the CabacCost_t
 * tables) and everything it produced (quantised coefficients, reconstructed coefficients, non-zero counts, coefficient
 * bits, the two distortions, yCbf, yDc).  Only the configuration the BASELINE presets use is recorded: no RDOQ/PM-core,
 * no spatial-SSE full loop, no CABAC-context update.  tests/golden/make_fullloop_golden.py builds the fixtures.
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

void __real_ProductFullLoop(EbPictureBufferDesc_t *inputPicturePtr, EB_U32 inputOriginIndex,
                            ModeDecisionCandidateBuffer_t *candidateBuffer, ModeDecisionContext_t *contextPtr,
                            const CodedUnitStats_t *cuStatsPtr, PictureControlSet_t *pictureControlSetPtr, EB_U32 qp,
                            EB_U32 *yCountNonZeroCoeffs, EB_U64 *yCoeffBits, EB_U64 *yFullDistortion);

#define FL_DUMP_MAGIC 0x4c4c5546U /* "FULL" */

typedef struct FullLoopRecord {
    uint32_t magic, record_size;
    uint64_t picture_number;
    uint32_t size, origin_x, origin_y, qp, slice_type, temporal_layer, pf_mode, cand_type, intra_luma_mode, full_lambda;
    uint32_t cbf_bits[4];        /* lumaCbfBits[0], [1], [5], [6]: zero / non-zero cbf at ctx 0 / 1 */
    uint32_t ycbf_before, ycbf_after;
    uint32_t nz_in[5], nz_out[5];
    uint64_t bits_in, bits_out, dist_in[2], dist_out[2];
    int16_t ydc[4];
    uint16_t cand_nz[4];
    SvtAmdCabacCost cost;
    int16_t residual[64 * 64], quant[64 * 64], recon[64 * 64]; /* size x size used, row pitch = size */
    uint32_t cabac_update, pad;  /* contextPtr->coeffCabacUpdate */
    uint32_t ctx_in[136], ctx_out[136]; /* candidateBuffer->candBuffCoeffCtxModel before / after (CoeffCtxtMdl_t) */
} FullLoopRecord;
_Static_assert(sizeof(CoeffCtxtMdl_t) == 136 * 4, "CoeffCtxtMdl_t");

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static FILE *g_file;
static int g_state, g_stride = 53;
static unsigned long g_calls;

static void grab(int16_t *dst, const EbPictureBufferDesc_t *pic, uint32_t origin, uint32_t size)
{
    const int16_t *src = (const int16_t *)pic->bufferY + origin;
    for (uint32_t y = 0; y < size; y++)
        memcpy(dst + y * size, src + (size_t)y * 64, size * sizeof(int16_t));
}

void __wrap_ProductFullLoop(EbPictureBufferDesc_t *inputPicturePtr, EB_U32 inputOriginIndex,
                            ModeDecisionCandidateBuffer_t *candidateBuffer, ModeDecisionContext_t *contextPtr,
                            const CodedUnitStats_t *cuStatsPtr, PictureControlSet_t *pcs, EB_U32 qp,
                            EB_U32 *yCountNonZeroCoeffs, EB_U64 *yCoeffBits, EB_U64 *yFullDistortion)
{
    if (g_state == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_state == 0) {
            const char *path = getenv("SVT_REF_FULLLOOP_DUMP"), *st = getenv("SVT_REF_FULLLOOP_STRIDE");
            g_file = path ? fopen(path, "wb") : NULL;
            if (st && atoi(st) > 0)
                g_stride = atoi(st);
            g_state = g_file ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
    int take = 0;
    if (g_state > 0 && (!contextPtr->rdoqPmCoreMethod || contextPtr->rdoqPmCoreMethod == EB_PMCORE) && !contextPtr->spatialSseFullLoop) {
        pthread_mutex_lock(&g_lock);
        take = (g_calls++ % (unsigned long)g_stride) == 0;
        pthread_mutex_unlock(&g_lock);
    }
    if (!take) {
        __real_ProductFullLoop(inputPicturePtr, inputOriginIndex, candidateBuffer, contextPtr, cuStatsPtr, pcs, qp,
                               yCountNonZeroCoeffs, yCoeffBits, yFullDistortion);
        return;
    }
    FullLoopRecord *r = (FullLoopRecord *)calloc(1, sizeof(*r));
    const uint32_t size = cuStatsPtr->size;
    const uint32_t origin = size == 64 ? 0 : cuStatsPtr->originX + (cuStatsPtr->originY << 6);
    const ModeDecisionCandidate_t *c = candidateBuffer->candidatePtr;
    r->magic = FL_DUMP_MAGIC, r->record_size = (uint32_t)sizeof(*r);
    r->picture_number = pcs->pictureNumber;
    r->size = size, r->origin_x = cuStatsPtr->originX, r->origin_y = cuStatsPtr->originY, r->qp = qp;
    r->slice_type = pcs->sliceType, r->temporal_layer = pcs->temporalLayerIndex;
    r->pf_mode = contextPtr->pfMdMode | ((uint32_t)contextPtr->rdoqPmCoreMethod << 16) /* = SvtAmdFullLoopIn.pf_mode + .pm_core */, r->cand_type = c->type, r->intra_luma_mode = c->intraLumaMode;
    r->full_lambda = contextPtr->fullLambda;
    r->cbf_bits[0] = c->mdRateEstimationPtr->lumaCbfBits[0], r->cbf_bits[1] = c->mdRateEstimationPtr->lumaCbfBits[1];
    r->cbf_bits[2] = c->mdRateEstimationPtr->lumaCbfBits[5], r->cbf_bits[3] = c->mdRateEstimationPtr->lumaCbfBits[6];
    r->ycbf_before = c->yCbf;
    memcpy(r->nz_in, yCountNonZeroCoeffs, sizeof(r->nz_in));
    r->bits_in = *yCoeffBits, r->dist_in[0] = yFullDistortion[0], r->dist_in[1] = yFullDistortion[1];
    memcpy(&r->cost, contextPtr->CabacCost, sizeof(r->cost));
    grab(r->residual, candidateBuffer->residualQuantCoeffPtr, origin, size);
    r->cabac_update = contextPtr->coeffCabacUpdate;
    memcpy(r->ctx_in, &candidateBuffer->candBuffCoeffCtxModel, sizeof(r->ctx_in));

    __real_ProductFullLoop(inputPicturePtr, inputOriginIndex, candidateBuffer, contextPtr, cuStatsPtr, pcs, qp,
                           yCountNonZeroCoeffs, yCoeffBits, yFullDistortion);

    r->ycbf_after = c->yCbf;
    memcpy(r->nz_out, yCountNonZeroCoeffs, sizeof(r->nz_out));
    r->bits_out = *yCoeffBits, r->dist_out[0] = yFullDistortion[0], r->dist_out[1] = yFullDistortion[1];
    memcpy(r->ydc, candidateBuffer->yDc, sizeof(r->ydc));
    memcpy(r->cand_nz, candidateBuffer->yCountNonZeroCoeffs, sizeof(r->cand_nz));
    memcpy(r->ctx_out, &candidateBuffer->candBuffCoeffCtxModel, sizeof(r->ctx_out));
    grab(r->quant, candidateBuffer->residualQuantCoeffPtr, origin, size);
    grab(r->recon, candidateBuffer->reconCoeffPtr, origin, size);
    pthread_mutex_lock(&g_lock);
    fwrite(r, sizeof(*r), 1, g_file);
    fflush(g_file);
    pthread_mutex_unlock(&g_lock);
    free(r);
}
