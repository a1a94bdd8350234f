/*
 * TEST INFRASTRUCTURE ONLY (oracle/): link-time interposer around the REFERENCE's EncodePassInterPrediction16bit
 * (Codec/EbInterPrediction.c:928, called per prediction unit of a 10-bit encode from EbCodingLoop.c).  Compiled only into
 * oracle/_ref/libsvtref.so with -Wl,--wrap=EncodePassInterPrediction16bit.  The 16-bit twin of ref_harness_inter_dump.c:
 *
 * With SVT_REF_INTER16_DUMP=<file>, the file receives two kinds of records:
 *   'P' the first time a (16-bit reference picture buffer, reference POC) pair is seen: the whole padded picture, three
 *       planes of 16-bit samples;
 *   'U' for a sample of the calls (every SVT_REF_INTER16_STRIDE-th, default 5): motion vectors, direction, unit geometry,
 *       the ids of the pictures it reads, and the three predicted blocks (16-bit samples) it left.
 * tests/golden/make_inter_golden.py builds the fixtures.  4:2:0 only.  No reference source here.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbReferenceObject.h"
#include "EbInterPrediction.h"

EB_ERRORTYPE __real_EncodePassInterPrediction16bit(MvUnit_t *mvUnit, EB_U16 puOriginX, EB_U16 puOriginY, EB_U8 puWidth, EB_U8 puHeight,
                                                   PictureControlSet_t *pcs, EbPictureBufferDesc_t *predictionPtr,
                                                   MotionCompensationPredictionContext_t *mcpContext);

#define INTER16_PIC_MAGIC 0x36495049U  /* "IPI6" */
#define INTER16_UNIT_MAGIC 0x364e5549U /* "IUN6" */
typedef struct Inter16PicHeader {
    uint32_t magic, id;
    uint32_t strideY, strideC, originX, originY, width, height, rowsY, rowsC; /* followed by (rowsY*strideY + 2*rowsC*strideC) u16 */
} Inter16PicHeader;
typedef struct Inter16UnitRecord {
    uint32_t magic, record_size;
    int16_t mv[2][2];
    uint16_t pu_x, pu_y;
    uint8_t pu_w, pu_h, pred_dir, pad;
    int32_t ref_id[2];
    uint16_t pred_y[64 * 64], pred_cb[32 * 32], pred_cr[32 * 32]; /* pu_w x pu_h (chroma halves), row pitch = that width */
} Inter16UnitRecord;

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static FILE *g_file;
static int g_state, g_stride = 5;
static unsigned long g_calls;
#define MAX_PICS 64
static struct { const void *buf; uint64_t poc; } g_pics[MAX_PICS];
static int g_npics;

/* must hold g_lock */
static int picture_id(const EbPictureBufferDesc_t *p, uint64_t poc)
{
    for (int i = 0; i < g_npics; i++)
        if (g_pics[i].buf == p->bufferY && g_pics[i].poc == poc)
            return i;
    if (g_npics == MAX_PICS)
        return -1;
    const int id = g_npics++;
    g_pics[id].buf = p->bufferY, g_pics[id].poc = poc;
    Inter16PicHeader h = {INTER16_PIC_MAGIC, (uint32_t)id, p->strideY, p->strideCb, p->originX, p->originY, p->width, p->height,
                          (uint32_t)(p->height + 2 * p->originY), (uint32_t)((p->height + 2 * p->originY) >> 1)};
    fwrite(&h, sizeof(h), 1, g_file);
    fwrite(p->bufferY, 2, (size_t)h.rowsY * h.strideY, g_file);
    fwrite(p->bufferCb, 2, (size_t)h.rowsC * h.strideC, g_file);
    fwrite(p->bufferCr, 2, (size_t)h.rowsC * h.strideC, g_file);
    return id;
}

EB_ERRORTYPE __wrap_EncodePassInterPrediction16bit(MvUnit_t *mvUnit, EB_U16 puOriginX, EB_U16 puOriginY, EB_U8 puWidth, EB_U8 puHeight,
                                                   PictureControlSet_t *pcs, EbPictureBufferDesc_t *predictionPtr,
                                                   MotionCompensationPredictionContext_t *mcpContext)
{
    if (g_state == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_state == 0) {
            const char *path = getenv("SVT_REF_INTER16_DUMP"), *st = getenv("SVT_REF_INTER16_STRIDE");
            g_file = path ? fopen(path, "wb") : NULL;
            if (st && atoi(st) > 0)
                g_stride = atoi(st);
            g_state = g_file ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
    const EB_ERRORTYPE rc = __real_EncodePassInterPrediction16bit(mvUnit, puOriginX, puOriginY, puWidth, puHeight, pcs, predictionPtr, mcpContext);
    if (g_state < 0 || predictionPtr->colorFormat != EB_YUV420 || puWidth > 64 || puHeight > 64)
        return rc;
    pthread_mutex_lock(&g_lock);
    if ((g_calls++ % (unsigned long)g_stride) == 0) {
        Inter16UnitRecord r;
        memset(&r, 0, sizeof(r));
        r.magic = INTER16_UNIT_MAGIC, r.record_size = (uint32_t)sizeof(r);
        r.pu_x = puOriginX, r.pu_y = puOriginY, r.pu_w = puWidth, r.pu_h = puHeight, r.pred_dir = mvUnit->predDirection;
        r.ref_id[0] = r.ref_id[1] = -1;
        for (int l = 0; l < 2; l++) {
            r.mv[l][0] = mvUnit->mv[l].x, r.mv[l][1] = mvUnit->mv[l].y;
            if (mvUnit->predDirection == l || mvUnit->predDirection == BI_PRED) {
                const EbReferenceObject_t *ro = (const EbReferenceObject_t *)pcs->refPicPtrArray[l]->objectPtr;
                r.ref_id[l] = picture_id(ro->referencePicture16bit, ro->refPOC);
            }
        }
        const uint32_t oy = (predictionPtr->originY + puOriginY) * predictionPtr->strideY + predictionPtr->originX + puOriginX;
        const uint32_t oc = (((predictionPtr->originY + puOriginY) * predictionPtr->strideCb) >> 1) + ((predictionPtr->originX + puOriginX) >> 1);
        const uint16_t *py = (const uint16_t *)predictionPtr->bufferY, *pcb = (const uint16_t *)predictionPtr->bufferCb,
                       *pcr = (const uint16_t *)predictionPtr->bufferCr;
        for (uint32_t y = 0; y < puHeight; y++)
            memcpy(r.pred_y + y * puWidth, py + oy + y * predictionPtr->strideY, 2 * puWidth);
        for (uint32_t y = 0; y < (uint32_t)(puHeight >> 1); y++) {
            memcpy(r.pred_cb + y * (puWidth >> 1), pcb + oc + y * predictionPtr->strideCb, puWidth);
            memcpy(r.pred_cr + y * (puWidth >> 1), pcr + oc + y * predictionPtr->strideCr, puWidth);
        }
        if (r.ref_id[0] != -1 || r.ref_id[1] != -1) {
            fwrite(&r, sizeof(r), 1, g_file);
            fflush(g_file);
        }
    }
    pthread_mutex_unlock(&g_lock);
    return rc;
}
