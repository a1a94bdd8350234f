/*
 * TEST INFRASTRUCTURE ONLY (oracle/): link-time interposers around the REFERENCE's SaoGenerationDecision /
 * SaoGenerationDecision16bit (Codec/EbSampleAdaptiveOffsetGenerationDecision.c:647, :936; called per LCU from
 * EbCodingLoop.c:4711, :4732).  Compiled only into oracle/_ref/libsvtref.so with -Wl,--wrap=... for the two symbols.
 *
 * With SVT_REF_SAODEC_DUMP=<file>, a sample of the calls (every SVT_REF_SAODEC_STRIDE-th, default 3) leaves one binary
 * record: the statistics the call gathered (whatever the arrays hold afterwards - the reduced modes refresh only part of
 * them), the lambdas, mode switches, rate tables, the left / upper neighbours' parameters, and the parameters and the two
 * costs it decided.  tests/golden/make_saodec_golden.py builds the fixtures.  No reference source here.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbCodingUnit.h"
#include "EbMdRateEstimation.h"
#include "EbSampleAdaptiveOffset.h"

EB_ERRORTYPE __real_SaoGenerationDecision(SaoStats_t *saoStats, SaoParameters_t *saoParams, MdRateEstimationContext_t *md, EB_U64 fullLambda,
                                          EB_U64 fullChromaLambdaSao, EB_BOOL mmSao, PictureControlSet_t *pcs, EB_U32 tbOriginX,
                                          EB_U32 tbOriginY, EB_U32 lcuWidth, EB_U32 lcuHeight, SaoParameters_t *saoPtr,
                                          SaoParameters_t *leftSaoPtr, SaoParameters_t *upSaoPtr, EB_S64 *saoLumaBestCost,
                                          EB_S64 *saoChromaBestCost);
EB_ERRORTYPE __real_SaoGenerationDecision16bit(EbPictureBufferDesc_t *inputLcuPtr, SaoStats_t *saoStats, SaoParameters_t *saoParams,
                                               MdRateEstimationContext_t *md, EB_U64 fullLambda, EB_U64 fullChromaLambdaSao, EB_BOOL mmSao,
                                               PictureControlSet_t *pcs, EB_U32 tbOriginX, EB_U32 tbOriginY, EB_U32 lcuWidth,
                                               EB_U32 lcuHeight, SaoParameters_t *saoPtr, SaoParameters_t *leftSaoPtr,
                                               SaoParameters_t *upSaoPtr, EB_S64 *saoLumaBestCost, EB_S64 *saoChromaBestCost);

#define SAODEC_DUMP_MAGIC 0x44414f53U /* "SOAD" */
typedef struct SaoParamsRec { uint8_t merge_left, merge_up, pad[2]; uint32_t type[2]; int32_t offset[3][4]; uint32_t band[3]; } SaoParamsRec;
typedef struct SaoDecRecord {
    uint32_t magic, record_size;
    uint32_t is16, mm_sao, temporal_layer, has_left, has_up, pad;
    uint64_t picture_number;
    uint32_t origin_x, origin_y;
    uint64_t lambda, chroma_lambda;
    uint32_t type_bits[6], merge_bits[2], offset_bits[8];
    SaoParamsRec left, up, out;
    int64_t luma_cost, chroma_cost;
    int32_t bo_diff[3][32];
    uint16_t bo_count[3][32];
    int32_t eo_diff[3][4][5];
    uint16_t eo_count[3][4][5];
} SaoDecRecord;

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static FILE *g_file;
static int g_state, g_stride = 3;
static unsigned long g_calls;

static void put_params(SaoParamsRec *d, const SaoParameters_t *s)
{
    d->merge_left = s->saoMergeLeftFlag, d->merge_up = s->saoMergeUpFlag;
    d->type[0] = s->saoTypeIndex[0], d->type[1] = s->saoTypeIndex[1];
    memcpy(d->offset, s->saoOffset, sizeof(d->offset));
    memcpy(d->band, s->saoBandPosition, sizeof(d->band));
}

static int want_record(void)
{
    if (g_state == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_state == 0) {
            const char *path = getenv("SVT_REF_SAODEC_DUMP"), *st = getenv("SVT_REF_SAODEC_STRIDE");
            g_file = path ? fopen(path, "wb") : NULL;
            if (st && atoi(st) > 0)
                g_stride = atoi(st);
            g_state = g_file ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
    if (g_state < 0)
        return 0;
    pthread_mutex_lock(&g_lock);
    const int take = (g_calls++ % (unsigned long)g_stride) == 0;
    pthread_mutex_unlock(&g_lock);
    return take;
}

static void record(int is16, const SaoStats_t *st, const MdRateEstimationContext_t *md, EB_U64 lambda, EB_U64 clambda, EB_BOOL mm,
                   const PictureControlSet_t *pcs, EB_U32 ox, EB_U32 oy, const SaoParameters_t *left, const SaoParameters_t *up, const SaoParameters_t *out,
                   EB_S64 lumaCost, EB_S64 chromaCost, const SaoParamsRec *leftBefore, const SaoParamsRec *upBefore)
{
    SaoDecRecord r;
    memset(&r, 0, sizeof(r));
    r.magic = SAODEC_DUMP_MAGIC, r.record_size = (uint32_t)sizeof(r), r.is16 = (uint32_t)is16, r.mm_sao = mm;
    r.temporal_layer = pcs->temporalLayerIndex, r.has_left = left != NULL, r.has_up = up != NULL;
    r.picture_number = pcs->pictureNumber, r.origin_x = ox, r.origin_y = oy;
    r.lambda = lambda, r.chroma_lambda = clambda;
    for (int k = 0; k < 6; k++)
        r.type_bits[k] = md->saoTypeIndexBits[k];
    for (int k = 0; k < 2; k++)
        r.merge_bits[k] = md->saoMergeFlagBits[k];
    for (int k = 0; k < 8; k++)
        r.offset_bits[k] = md->saoOffsetTrunUnaryBits[k];
    r.left = *leftBefore, r.up = *upBefore;
    put_params(&r.out, out);
    r.luma_cost = lumaCost, r.chroma_cost = chromaCost;
    for (int c = 0; c < 3; c++) {
        memcpy(r.bo_diff[c], st->boDiff[c], sizeof(r.bo_diff[c]));
        memcpy(r.bo_count[c], st->boCount[c], sizeof(r.bo_count[c]));
    }
    memcpy(r.eo_diff, st->eoDiff, sizeof(r.eo_diff));
    memcpy(r.eo_count, st->eoCount, sizeof(r.eo_count));
    pthread_mutex_lock(&g_lock);
    fwrite(&r, sizeof(r), 1, g_file);
    fflush(g_file);
    pthread_mutex_unlock(&g_lock);
}

EB_ERRORTYPE __wrap_SaoGenerationDecision(SaoStats_t *saoStats, SaoParameters_t *saoParams, MdRateEstimationContext_t *md, EB_U64 fullLambda,
                                          EB_U64 fullChromaLambdaSao, EB_BOOL mmSao, PictureControlSet_t *pcs, EB_U32 tbOriginX,
                                          EB_U32 tbOriginY, EB_U32 lcuWidth, EB_U32 lcuHeight, SaoParameters_t *saoPtr,
                                          SaoParameters_t *leftSaoPtr, SaoParameters_t *upSaoPtr, EB_S64 *saoLumaBestCost,
                                          EB_S64 *saoChromaBestCost)
{
    const int take = want_record();
    SaoParamsRec lb, ub;
    memset(&lb, 0, sizeof(lb)), memset(&ub, 0, sizeof(ub));
    if (take && leftSaoPtr)
        put_params(&lb, leftSaoPtr);
    if (take && upSaoPtr)
        put_params(&ub, upSaoPtr);
    const EB_ERRORTYPE rc = __real_SaoGenerationDecision(saoStats, saoParams, md, fullLambda, fullChromaLambdaSao, mmSao, pcs, tbOriginX,
                                                         tbOriginY, lcuWidth, lcuHeight, saoPtr, leftSaoPtr, upSaoPtr, saoLumaBestCost,
                                                         saoChromaBestCost);
    if (take && saoParams == saoPtr)
        record(0, saoStats, md, fullLambda, fullChromaLambdaSao, mmSao, pcs, tbOriginX, tbOriginY, leftSaoPtr, upSaoPtr, saoPtr, *saoLumaBestCost, *saoChromaBestCost, &lb, &ub);
    return rc;
}

EB_ERRORTYPE __wrap_SaoGenerationDecision16bit(EbPictureBufferDesc_t *inputLcuPtr, SaoStats_t *saoStats, SaoParameters_t *saoParams,
                                               MdRateEstimationContext_t *md, EB_U64 fullLambda, EB_U64 fullChromaLambdaSao, EB_BOOL mmSao,
                                               PictureControlSet_t *pcs, EB_U32 tbOriginX, EB_U32 tbOriginY, EB_U32 lcuWidth,
                                               EB_U32 lcuHeight, SaoParameters_t *saoPtr, SaoParameters_t *leftSaoPtr,
                                               SaoParameters_t *upSaoPtr, EB_S64 *saoLumaBestCost, EB_S64 *saoChromaBestCost)
{
    const int take = want_record();
    SaoParamsRec lb, ub;
    memset(&lb, 0, sizeof(lb)), memset(&ub, 0, sizeof(ub));
    if (take && leftSaoPtr)
        put_params(&lb, leftSaoPtr);
    if (take && upSaoPtr)
        put_params(&ub, upSaoPtr);
    const EB_ERRORTYPE rc = __real_SaoGenerationDecision16bit(inputLcuPtr, saoStats, saoParams, md, fullLambda, fullChromaLambdaSao, mmSao, pcs,
                                                              tbOriginX, tbOriginY, lcuWidth, lcuHeight, saoPtr, leftSaoPtr, upSaoPtr,
                                                              saoLumaBestCost, saoChromaBestCost);
    if (take && saoParams == saoPtr)
        record(1, saoStats, md, fullLambda, fullChromaLambdaSao, mmSao, pcs, tbOriginX, tbOriginY, leftSaoPtr, upSaoPtr, saoPtr, *saoLumaBestCost, *saoChromaBestCost, &lb, &ub);
    return rc;
}
