/*
 * TEST INFRASTRUCTURE ONLY (oracle/): link-time interposer around the
 * REFERENCE's MotionEstimateLcu (Codec/EbMotionEstimation.c:3671), compiled
 * only into oracle/_ref/libsvtref.so with -Wl,--wrap=MotionEstimateLcu.
 *
 * When the environment variable SVT_REF_ME_DUMP names a file, every call of
 * the real function is followed by one binary record holding
 *   - the picture-level controls the call read (as an SvtAmdMeParams),
 *   - the identity of the picture, of its references and of the LCU,
 *   - everything the call produced (meResults[lcu][85], pLcuBestSad/MV, search
 *     area origins).
 * tests/golden/make_me_golden.py turns such dumps into the committed golden
 * fixtures that pin oracle/svt_oracle_me.c (and through it the HIP path).
 * Without the variable the wrapper is a tail call.
 *
 * This file contains no reference source; it includes the reference headers
 * only to read its structs.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbMotionEstimation.h"
#include "EbMotionEstimationContext.h"
#include "EbReferenceObject.h"

#include "../include/svt_hevc_amd.h"

EB_ERRORTYPE __real_MotionEstimateLcu(PictureParentControlSet_t *pcs, EB_U32 lcuIndex, EB_U32 lcuOriginX,
                                      EB_U32 lcuOriginY, MeContext_t *ctx, EbPictureBufferDesc_t *inputPtr);

uint64_t svt_ref_front_time_begin(void);          /* ref_harness_front_time.c */
void svt_ref_front_time_end(int which, uint64_t t0);

#define DUMP_MAGIC 0x4d45444dU /* "MDEM" */

typedef struct MeDumpRecord {
    uint32_t magic;
    uint32_t record_size;
    uint64_t picture_number;
    uint64_t ref_poc[2];
    uint32_t lcu_index;
    uint32_t lcu_origin_x, lcu_origin_y;
    uint32_t slice_type;
    uint32_t enc_mode;
    uint32_t luma_crc;          /* additive checksum of the padded-input luma interior */
    uint32_t ref_crc[2];        /* same for the two references                          */
    SvtAmdMeParams params;
    SvtAmdMeLcuResult result;
} MeDumpRecord;

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static FILE *g_file;
static int g_state; /* 0 unknown, 1 active, -1 off */

static uint32_t plane_checksum(const EbPictureBufferDesc_t *p)
{
    uint32_t s = 0;
    for (uint32_t y = 0; y < p->height; y++) {
        const uint8_t *row = p->bufferY + (size_t)(p->originY + y) * p->strideY + p->originX;
        for (uint32_t x = 0; x < p->width; x++)
            s = s * 31u + row[x];
    }
    return s;
}

/* SVT_REF_PA_DUMP: what the picture-analysis process left for this LCU / picture before motion estimation (GatheringPictureStatistics,
 * Codec/EbPictureAnalysisProcess.c:3995): variance[lcu][85] / yMean[lcu][85] (ComputeBlockMeanComputeVariance :1646) per call, and with LCU 0 the
 * luma histograms of the regions (SubSampleLumaGeneratePixelIntensityHistogramBins :3384).  tests/golden/make_pa_golden.py */
#define PA_MAGIC 0x50414453U
typedef struct PaDumpRecord {
    uint32_t magic, kind;        /* kind 0: LCU record, 1: picture record */
    uint64_t picture_number;
    uint32_t lcu_index, regions_w, regions_h, pad;
    uint16_t variance[85];
    uint8_t y_mean[85];
    uint8_t average_intensity;
    uint32_t histogram[4][4][256];
    uint8_t region_average[4][4];
} PaDumpRecord;
static FILE *g_pa_file;
static int g_pa_state;
static void pa_dump(PictureParentControlSet_t *pcs, EB_U32 lcuIndex)
{
    if (g_pa_state == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_pa_state == 0) {
            const char *path = getenv("SVT_REF_PA_DUMP");
            g_pa_file = path ? fopen(path, "wb") : NULL;
            g_pa_state = g_pa_file ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
    if (g_pa_state < 0)
        return;
    SequenceControlSet_t *scs = (SequenceControlSet_t *)pcs->sequenceControlSetWrapperPtr->objectPtr;
    PaDumpRecord *r = (PaDumpRecord *)calloc(1, sizeof(*r));
    if (!r)
        return;
    r->magic = PA_MAGIC, r->picture_number = pcs->pictureNumber, r->lcu_index = lcuIndex;
    memcpy(r->variance, pcs->variance[lcuIndex], sizeof(r->variance));
    memcpy(r->y_mean, pcs->yMean[lcuIndex], sizeof(r->y_mean));
    if (lcuIndex == 0) {
        r->kind = 1;
        r->regions_w = scs->pictureAnalysisNumberOfRegionsPerWidth, r->regions_h = scs->pictureAnalysisNumberOfRegionsPerHeight;
        r->average_intensity = pcs->averageIntensity[0];
        for (uint32_t a = 0; a < r->regions_w && a < 4; a++)
            for (uint32_t b = 0; b < r->regions_h && b < 4; b++) {
                memcpy(r->histogram[a][b], pcs->pictureHistogram[a][b][0], sizeof(r->histogram[a][b]));
                r->region_average[a][b] = (uint8_t)pcs->averageIntensityPerRegion[a][b][0];
            }
    }
    pthread_mutex_lock(&g_lock);
    fwrite(r, sizeof(*r), 1, g_pa_file);
    fflush(g_pa_file);
    pthread_mutex_unlock(&g_lock);
    free(r);
}

EB_ERRORTYPE __wrap_MotionEstimateLcu(PictureParentControlSet_t *pcs, EB_U32 lcuIndex, EB_U32 lcuOriginX,
                                      EB_U32 lcuOriginY, MeContext_t *ctx, EbPictureBufferDesc_t *inputPtr)
{
    const uint64_t t0 = svt_ref_front_time_begin();
    EB_ERRORTYPE err = __real_MotionEstimateLcu(pcs, lcuIndex, lcuOriginX, lcuOriginY, ctx, inputPtr);
    svt_ref_front_time_end(0, t0);
    if (g_state == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_state == 0) {
            const char *path = getenv("SVT_REF_ME_DUMP");
            g_file = path ? fopen(path, "wb") : NULL;
            g_state = g_file ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
    pa_dump(pcs, lcuIndex);
    if (g_state < 0)
        return err;

    SequenceControlSet_t *scs = (SequenceControlSet_t *)pcs->sequenceControlSetWrapperPtr->objectPtr;
    MeDumpRecord *r = (MeDumpRecord *)calloc(1, sizeof(*r));
    if (!r)
        return err;
    r->magic = DUMP_MAGIC;
    r->record_size = (uint32_t)sizeof(*r);
    r->picture_number = pcs->pictureNumber;
    r->lcu_index = lcuIndex;
    r->lcu_origin_x = lcuOriginX;
    r->lcu_origin_y = lcuOriginY;
    r->slice_type = pcs->sliceType;
    r->enc_mode = pcs->encMode;
    const int nlists = (pcs->sliceType == EB_P_PICTURE) ? 1 : 2;
    for (int l = 0; l < nlists; l++)
        r->ref_poc[l] = pcs->refPicPocArray[l];
    if (lcuIndex == 0) { /* once per picture is enough for the input-identity check */
        EbPaReferenceObject_t *cur = (EbPaReferenceObject_t *)pcs->paReferencePictureWrapperPtr->objectPtr;
        r->luma_crc = plane_checksum(cur->inputPaddedPicturePtr);
        for (int l = 0; l < nlists; l++) {
            EbPaReferenceObject_t *ro = (EbPaReferenceObject_t *)pcs->refPaPicPtrArray[l]->objectPtr;
            r->ref_crc[l] = plane_checksum(ro->inputPaddedPicturePtr);
        }
    }

    SvtAmdMeParams *p = &r->params;
    p->luma_width = scs->lumaWidth;
    p->luma_height = scs->lumaHeight;
    p->num_lists = (uint8_t)nlists;
    p->temporal_layer_index = pcs->temporalLayerIndex;
    p->ref_pocs_equal = (nlists == 2 && pcs->refPicPocArray[0] == pcs->refPicPocArray[1]);
    p->enable_hme_flag = pcs->enableHmeFlag;
    p->enable_hme_level0 = pcs->enableHmeLevel0Flag;
    p->enable_hme_level1 = pcs->enableHmeLevel1Flag;
    p->enable_hme_level2 = pcs->enableHmeLevel2Flag;
    p->one_quadrant_hme = ctx->oneQuadrantHME;
    p->update_hme_search_center = ctx->updateHmeSearchCenter;
    p->num_hme_regions_w = (uint8_t)ctx->numberHmeSearchRegionInWidth;
    p->num_hme_regions_h = (uint8_t)ctx->numberHmeSearchRegionInHeight;
    p->search_area_width = (uint8_t)ctx->searchAreaWidth;
    p->search_area_height = (uint8_t)ctx->searchAreaHeight;
    p->fractional_search_method = ctx->fractionalSearchMethod;
    p->fractional_search_model = ctx->fractionalSearchModel;
    p->fractional_search_64x64 = ctx->fractionalSearch64x64;
    p->cu8x8_mode = pcs->cu8x8Mode;
    p->cu16x16_mode = pcs->cu16x16Mode;
    p->hme_l0_total_w = ctx->hmeLevel0TotalSearchAreaWidth;
    p->hme_l0_total_h = ctx->hmeLevel0TotalSearchAreaHeight;
    for (int k = 0; k < 2; k++) {
        p->hme_l0_w[k] = ctx->hmeLevel0SearchAreaInWidthArray[k];
        p->hme_l0_h[k] = ctx->hmeLevel0SearchAreaInHeightArray[k];
        p->hme_l1_w[k] = ctx->hmeLevel1SearchAreaInWidthArray[k];
        p->hme_l1_h[k] = ctx->hmeLevel1SearchAreaInHeightArray[k];
        p->hme_l2_w[k] = ctx->hmeLevel2SearchAreaInWidthArray[k];
        p->hme_l2_h[k] = ctx->hmeLevel2SearchAreaInHeightArray[k];
    }
    p->hme_l0_mult_x = (uint16_t)HME_LEVEL_0_SEARCH_AREA_MULTIPLIER_X[pcs->hierarchicalLevels][pcs->temporalLayerIndex];
    p->hme_l0_mult_y = (uint16_t)HME_LEVEL_0_SEARCH_AREA_MULTIPLIER_Y[pcs->hierarchicalLevels][pcs->temporalLayerIndex];
    p->lambda = (uint32_t)ctx->lambda;
    for (int k = 0; k < 12; k++)
        p->mvd_bits[k] = ctx->mvdBitsArray[k];

    SvtAmdMeLcuResult *o = &r->result;
    for (int pu = 0; pu < SVT_AMD_ME_PU_COUNT; pu++) {
        const MeCuResults_t *m = &pcs->meResults[lcuIndex][pu];
        o->pu[pu].x_mv_l0 = m->xMvL0;
        o->pu[pu].y_mv_l0 = m->yMvL0;
        o->pu[pu].x_mv_l1 = m->xMvL1;
        o->pu[pu].y_mv_l1 = m->yMvL1;
        o->pu[pu].total_me_candidate_index = m->totalMeCandidateIndex;
        for (int k = 0; k < 3; k++) {
            o->pu[pu].distortion[k] = m->distortionDirection[k].distortion;
            o->pu[pu].direction[k] = (uint8_t)m->distortionDirection[k].direction;
        }
    }
    for (int l = 0; l < nlists; l++) {
        memcpy(o->best_sad[l], ctx->pLcuBestSad[l][0], sizeof(uint32_t) * SVT_AMD_ME_PU_COUNT);
        memcpy(o->best_mv[l], ctx->pLcuBestMV[l][0], sizeof(uint32_t) * SVT_AMD_ME_PU_COUNT);
        o->search_origin_x[l] = ctx->xSearchAreaOrigin[l][0];
        o->search_origin_y[l] = ctx->ySearchAreaOrigin[l][0];
    }

    pthread_mutex_lock(&g_lock);
    fwrite(r, sizeof(*r), 1, g_file);
    fflush(g_file);
    pthread_mutex_unlock(&g_lock);
    free(r);
    return err;
}
