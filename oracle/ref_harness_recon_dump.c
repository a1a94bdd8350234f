/*
 * TEST INFRASTRUCTURE ONLY (oracle/): interposer on the REFERENCE's EncodeGenerateRecon / EncodeGenerateRecon16bit
 * (Codec/EbCodingLoop.c:1084, :1660).  Both are `static`, but the encode pass reaches them only through the global table
 * EncodeGenerateReconFunctionPtr[2] (:1807), whose slots a constructor of this file swaps for recording wrappers.
 * Compiled only into oracle/_ref/libsvtref.so.
 *
 * With SVT_REF_RECON_DUMP=<file>, a sample of the calls (every SVT_REF_RECON_STRIDE-th, default 41) leaves one binary
 * record per plane the call reconstructs (cbf set, CU not skipped): transform size, DC-only / DST flags, bytes per sample,
 * the inverse-quantised coefficients and the prediction the call found, and the reconstruction it left.
 * tests/golden/make_recon_golden.py builds the fixtures.  No reference source here.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbEncDecProcess.h"
#include "EbCodingUnit.h"
#include "EbTransformUnit.h"

typedef void (*ReconFn)(EncDecContext_t *, EB_U32, EB_U32, EB_U32, EB_COLOR_FORMAT, EB_BOOL, EB_U32, EbPictureBufferDesc_t *,
                        EbPictureBufferDesc_t *, EB_S16 *);
extern ReconFn EncodeGenerateReconFunctionPtr[2];
static ReconFn g_real[2];

#define RECON_DUMP_MAGIC 0x4e4f4352U /* "RCON" */
typedef struct ReconRecord {
    uint32_t magic, record_size;
    uint32_t size, plane, only_dc, dst, bytes_per_sample, pad;
    int16_t coeff[32 * 32];
    uint16_t pred[32 * 32], recon[32 * 32]; /* size x size used, row pitch = size; 8-bit samples widened */
} ReconRecord;

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static FILE *g_file;
static int g_state, g_stride = 41;
static unsigned long g_calls;

static void grab_samples(uint16_t *dst, const EbPictureBufferDesc_t *pic, int plane, uint32_t off, uint32_t n, int bps)
{
    const uint8_t *base = plane == 0 ? pic->bufferY : plane == 1 ? pic->bufferCb : pic->bufferCr;
    const uint32_t stride = plane == 0 ? pic->strideY : plane == 1 ? pic->strideCb : pic->strideCr;
    for (uint32_t y = 0; y < n; y++)
        for (uint32_t x = 0; x < n; x++)
            dst[y * n + x] = bps == 1 ? base[off + y * stride + x] : ((const uint16_t *)base)[off + y * stride + x];
}

static void wrapper(int is16, EncDecContext_t *ctx, EB_U32 originX, EB_U32 originY, EB_U32 componentMask, EB_COLOR_FORMAT colorFormat,
                    EB_BOOL secondChroma, EB_U32 tuSize, EbPictureBufferDesc_t *predSamples, EbPictureBufferDesc_t *residual16bit,
                    EB_S16 *scratch)
{
    if (g_state == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_state == 0) {
            const char *path = getenv("SVT_REF_RECON_DUMP"), *st = getenv("SVT_REF_RECON_STRIDE");
            g_file = path ? fopen(path, "wb") : NULL;
            if (st && atoi(st) > 0)
                g_stride = atoi(st);
            g_state = g_file ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
    int take = 0;
    if (g_state > 0 && colorFormat == EB_YUV420 && !secondChroma) {
        pthread_mutex_lock(&g_lock);
        take = (g_calls++ % (unsigned long)g_stride) == 0;
        pthread_mutex_unlock(&g_lock);
    }
    ReconRecord *recs[3] = {NULL, NULL, NULL};
    uint32_t offs[3] = {0, 0, 0};
    if (take) {
        const CodingUnit_t *cu = ctx->cuPtr;
        const TransformUnit_t *tu = &cu->transformUnitArray[ctx->tuItr];
        const int bps = is16 ? 2 : 1;
        for (int p = 0; p < 3; p++) {
            const int wanted = p == 0 ? (componentMask & PICTURE_BUFFER_DESC_LUMA_MASK) != 0 : (componentMask & PICTURE_BUFFER_DESC_CHROMA_MASK) != 0;
            const int cbf = p == 0 ? tu->lumaCbf : p == 1 ? tu->cbCbf : tu->crCbf;
            if (!wanted || !cbf || cu->skipFlag)
                continue;
            const uint32_t n = p == 0 ? tuSize : (tuSize == 4 ? 4 : tuSize >> 1);
            ReconRecord *r = (ReconRecord *)calloc(1, sizeof(*r));
            r->magic = RECON_DUMP_MAGIC, r->record_size = (uint32_t)sizeof(*r), r->size = n, r->plane = (uint32_t)p, r->bytes_per_sample = (uint32_t)bps;
            r->dst = p == 0 && tuSize == 4;
            r->only_dc = tuSize == 4 ? 0 : (p == 0 ? (tu->transCoeffShapeLuma == ONLY_DC_SHAPE || tu->isOnlyDc[0])
                                                   : (tu->transCoeffShapeChroma == ONLY_DC_SHAPE || tu->isOnlyDc[p]));
            uint32_t scratchOff, cstride;
            if (p == 0) {
                offs[p] = (predSamples->originY + originY) * predSamples->strideY + (predSamples->originX + originX);
                scratchOff = ((originY & 63) * 64) + (originX & 63), cstride = 64;
            } else {
                const uint32_t stride = p == 1 ? predSamples->strideCb : predSamples->strideCr;
                offs[p] = ((predSamples->originX + originX) >> 1) + (((predSamples->originY + originY) >> 1) * stride);
                scratchOff = ((originX & 63) >> 1) + (((originY & 63) >> 1) * 32), cstride = 32;
            }
            const int16_t *c = (const int16_t *)(p == 0 ? residual16bit->bufferY : p == 1 ? residual16bit->bufferCb : residual16bit->bufferCr) + scratchOff;
            for (uint32_t y = 0; y < n; y++)
                memcpy(r->coeff + y * n, c + y * cstride, n * sizeof(int16_t));
            grab_samples(r->pred, predSamples, p, offs[p], n, bps);
            recs[p] = r;
        }
    }
    g_real[is16](ctx, originX, originY, componentMask, colorFormat, secondChroma, tuSize, predSamples, residual16bit, scratch);
    for (int p = 0; p < 3; p++) {
        if (!recs[p])
            continue;
        grab_samples(recs[p]->recon, predSamples, p, offs[p], recs[p]->size, (int)recs[p]->bytes_per_sample);
        pthread_mutex_lock(&g_lock);
        fwrite(recs[p], sizeof(ReconRecord), 1, g_file);
        fflush(g_file);
        pthread_mutex_unlock(&g_lock);
        free(recs[p]);
    }
}

static void wrap8(EncDecContext_t *a, EB_U32 b, EB_U32 c, EB_U32 d, EB_COLOR_FORMAT e, EB_BOOL f, EB_U32 g, EbPictureBufferDesc_t *h,
                  EbPictureBufferDesc_t *i, EB_S16 *j)
{
    wrapper(0, a, b, c, d, e, f, g, h, i, j);
}
static void wrap16(EncDecContext_t *a, EB_U32 b, EB_U32 c, EB_U32 d, EB_COLOR_FORMAT e, EB_BOOL f, EB_U32 g, EbPictureBufferDesc_t *h,
                   EbPictureBufferDesc_t *i, EB_S16 *j)
{
    wrapper(1, a, b, c, d, e, f, g, h, i, j);
}

__attribute__((constructor)) static void install(void)
{
    g_real[0] = EncodeGenerateReconFunctionPtr[0], g_real[1] = EncodeGenerateReconFunctionPtr[1];
    EncodeGenerateReconFunctionPtr[0] = wrap8, EncodeGenerateReconFunctionPtr[1] = wrap16;
}
