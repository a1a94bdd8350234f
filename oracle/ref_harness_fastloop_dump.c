/*
 * TEST INFRASTRUCTURE ONLY (oracle/): link-time interposers that catch the distortion stage of the REFERENCE's mode-decision
 * fast loop (ProductPerformFastLoop, Codec/EbProductCodingLoop.c:1911-2190): for every candidate of the second fast-cost
 * search the loop predicts (ProductMdFastPuPrediction -> the candidate's prediction function), measures the luma SAD and,
 * with useChromaInformationInFastLoop, the Cb + Cr SADs against the source block (NxMSadKernel_funcPtrArray, :2044-2078),
 * and hands both to the fast-cost function of the candidate type.  Compiled only into oracle/_ref/libsvtref.so with
 * -Wl,--wrap= for the four fast-cost functions (Intra2Nx2NFastCostIsliceOpt, Intra2Nx2NFastCostPsliceOpt,
 * InterFastCostPsliceOpt, InterFastCostBsliceOpt, Codec/EbRateDistortionCost.c) and the prediction functions that are not
 * wrapped elsewhere (Inter2Nx2NPuPredictionHevc, Inter2Nx2NPuPredictionInterpolationFree); the IntraPredictionCl / IntraPredictionOl
 * interposers of ref_harness_intra_dump.c report here through svt_ref_fastloop_note_prediction().
 *
 * A fast-cost call that directly follows a prediction call for the same candidate buffer on the same thread is a
 * second-loop call.  With SVT_REF_FASTLOOP_DUMP=<file>, every SVT_REF_FASTLOOP_STRIDE-th of them (default 29) leaves one
 * record: the source block, the predicted block (three planes), the two distortions handed over and the switches that
 * shape them.  tests/golden/make_fastloop_golden.py builds the fixtures.  No reference source here.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbModeDecisionProcess.h"
#include "EbModeDecision.h"
#include "EbRateDistortionCost.h"

#define FASTLOOP_DUMP_MAGIC 0x504c5346U /* "FSLP" */
typedef struct FastLoopRecord {
    uint32_t magic, record_size;
    uint32_t size, cand_type, slice_type, use_chroma, mpm_flag, distortion_ready, noise_lcu, intra_luma_mode;
    uint64_t luma_distortion, chroma_distortion, me_distortion;
    uint8_t src_y[64 * 64], src_cb[32 * 32], src_cr[32 * 32];   /* size x size (chroma halves), row pitch = that size */
    uint8_t pred_y[64 * 64], pred_cb[32 * 32], pred_cr[32 * 32];
} FastLoopRecord;

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static FILE *g_file;
static int g_state, g_stride = 29;
static unsigned long g_calls;
static __thread const void *t_predicted; /* candidate buffer the last prediction call of this thread filled */

void svt_ref_fastloop_note_prediction(const void *candidateBuffer) { t_predicted = candidateBuffer; }

static void maybe_record(ModeDecisionContext_t *md, ModeDecisionCandidateBuffer_t *cb, EB_U64 lumaDistortion, EB_U64 chromaDistortion,
                         PictureControlSet_t *pcs)
{
    const void *was = t_predicted;
    t_predicted = NULL;
    if (was != (const void *)cb)
        return;
    if (g_state == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_state == 0) {
            const char *path = getenv("SVT_REF_FASTLOOP_DUMP"), *st = getenv("SVT_REF_FASTLOOP_STRIDE");
            g_file = path ? fopen(path, "wb") : NULL;
            if (st && atoi(st) > 0)
                g_stride = atoi(st);
            g_state = g_file ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
    if (g_state < 0)
        return;
    pthread_mutex_lock(&g_lock);
    const int take = (g_calls++ % (unsigned long)g_stride) == 0;
    pthread_mutex_unlock(&g_lock);
    if (!take)
        return;
    FastLoopRecord *r = (FastLoopRecord *)calloc(1, sizeof(*r));
    const uint32_t size = md->cuStats->size, c = size >> 1, ox = md->cuOriginX, oy = md->cuOriginY;
    const ModeDecisionCandidate_t *cand = cb->candidatePtr;
    r->magic = FASTLOOP_DUMP_MAGIC, r->record_size = (uint32_t)sizeof(*r);
    r->size = size, r->cand_type = cand->type, r->slice_type = pcs->sliceType, r->use_chroma = md->useChromaInformationInFastLoop;
    r->mpm_flag = cand->mpmFlag, r->distortion_ready = cand->distortionReady, r->intra_luma_mode = cand->intraLumaMode;
    r->noise_lcu = pcs->ParentPcsPtr->cmplxStatusLcu[md->lcuPtr->index] == CMPLX_NOISE;
    r->luma_distortion = lumaDistortion, r->chroma_distortion = chromaDistortion, r->me_distortion = cand->meDistortion;
    const EbPictureBufferDesc_t *in = pcs->ParentPcsPtr->enhancedPicturePtr, *pr = cb->predictionPtr;
    const uint8_t *sy = in->bufferY + (in->originY + oy) * in->strideY + in->originX + ox;
    const uint8_t *scb = in->bufferCb + (((in->originY + oy) * in->strideCb) >> 1) + ((in->originX + ox) >> 1);
    const uint8_t *scr = in->bufferCr + (((in->originY + oy) * in->strideCr) >> 1) + ((in->originX + ox) >> 1);
    const uint32_t po = (oy & 63) * 64 + (ox & 63), pc = (((oy & 63) * 32) + (ox & 63)) >> 1;
    for (uint32_t y = 0; y < size; y++) {
        memcpy(r->src_y + y * size, sy + y * in->strideY, size);
        memcpy(r->pred_y + y * size, pr->bufferY + po + y * pr->strideY, size);
    }
    for (uint32_t y = 0; y < c; y++) {
        memcpy(r->src_cb + y * c, scb + y * in->strideCb, c);
        memcpy(r->src_cr + y * c, scr + y * in->strideCr, c);
        memcpy(r->pred_cb + y * c, pr->bufferCb + pc + y * pr->strideCb, c);
        memcpy(r->pred_cr + y * c, pr->bufferCr + pc + y * pr->strideCr, c);
    }
    pthread_mutex_lock(&g_lock);
    fwrite(r, sizeof(*r), 1, g_file);
    fflush(g_file);
    pthread_mutex_unlock(&g_lock);
    free(r);
}

#define COST_ARGS struct ModeDecisionContext_s *md, CodingUnit_t *cu, struct ModeDecisionCandidateBuffer_s *cb, EB_U32 qp, EB_U64 lumaD, \
                  EB_U64 chromaD, EB_U64 lambda, PictureControlSet_t *pcs
#define WRAP_COST(name)                                                  \
    EB_ERRORTYPE __real_##name(COST_ARGS);                               \
    EB_ERRORTYPE __wrap_##name(COST_ARGS)                                \
    {                                                                    \
        maybe_record(md, cb, lumaD, chromaD, pcs);                       \
        return __real_##name(md, cu, cb, qp, lumaD, chromaD, lambda, pcs); \
    }
WRAP_COST(Intra2Nx2NFastCostIsliceOpt)
WRAP_COST(Intra2Nx2NFastCostPsliceOpt)
WRAP_COST(InterFastCostPsliceOpt)
WRAP_COST(InterFastCostBsliceOpt)

#define PRED_ARGS ModeDecisionContext_t *md, EB_U32 mask, PictureControlSet_t *pcs, ModeDecisionCandidateBuffer_t *cb
#define WRAP_PRED(name)                                  \
    EB_ERRORTYPE __real_##name(PRED_ARGS);               \
    EB_ERRORTYPE __wrap_##name(PRED_ARGS)                \
    {                                                    \
        const EB_ERRORTYPE rc = __real_##name(md, mask, pcs, cb); \
        t_predicted = cb;                                \
        return rc;                                       \
    }
WRAP_PRED(Inter2Nx2NPuPredictionHevc)
WRAP_PRED(Inter2Nx2NPuPredictionInterpolationFree)
