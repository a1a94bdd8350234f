/*
 * TEST INFRASTRUCTURE ONLY (oracle/): link-time interposers around the REFERENCE's per-LCU deblocking drivers
 * LCUInternalAreaDLFCore(16bit) (Codec/EbDeblockingFilter.c:2222, :2556) and LCUPictureEdgeDLFCore(16bit) (:3518, :3923),
 * the first and the last of the three driver calls EncodePass makes per LCU (Codec/EbCodingLoop.c:4600-4631).  Compiled
 * only into oracle/_ref/libsvtref.so with -Wl,--wrap=... for the four symbols.
 *
 * With SVT_REF_DLF_DUMP=<file>, every picture leaves one binary record: the reconstructed picture BEFORE deblocking
 * (assembled from each LCU's block as its first driver call finds it - nothing has touched an LCU's samples before
 * that, since the left / upper neighbours' passes only write their own side of the LCU boundary after it), the two
 * boundary-strength arrays of every LCU, the picture's qpArray, the slice's tc / beta / chroma-qp offsets, and the
 * picture AFTER the last LCU's last driver call (= deblocked, SAO not yet applied).
 * tests/golden/make_dlf_golden.py builds the fixtures.  No reference source here.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbDeblockingFilter.h"
#include "EbCodingUnit.h"
#include "EbReferenceObject.h"
#include "EbUtility.h"

#define DLF_DUMP_MAGIC 0x20464c44U /* "DLF " */

typedef struct DlfRecordHeader {
    uint32_t magic, header_size;
    uint64_t picture_number;
    uint32_t width, height, bytes_per_sample, slice_type, lcu_cols, lcu_rows, qp_stride, qp_size;
    int32_t tc_offset, beta_offset, cb_qp_offset, cr_qp_offset;
    /* followed by: pre Y, Cb, Cr | post Y, Cb, Cr (tight planes) | bs_v[nlcu][256] | bs_h[nlcu][256] | qpArray */
} DlfRecordHeader;

typedef struct DlfPicture {
    PictureControlSet_t *pcs;
    uint64_t picture_number;
    uint32_t seen_first, seen_last, nlcu, bps;
    uint8_t *pre[3], *bsv, *bsh;
} DlfPicture;

#define MAX_INFLIGHT 64
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static DlfPicture g_pics[MAX_INFLIGHT];
static FILE *g_file;
static int g_state;

static int dump_on(void)
{
    if (g_state == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_state == 0) {
            const char *path = getenv("SVT_REF_DLF_DUMP");
            g_file = path ? fopen(path, "wb") : NULL;
            g_state = g_file ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
    return g_state > 0;
}

static void plane_geometry(const EbPictureBufferDesc_t *pic, const SequenceControlSet_t *scs, int plane, uint32_t *w,
                           uint32_t *h, uint32_t *stride, size_t *origin)
{
    const uint32_t sh = plane ? 1 : 0;
    *w = scs->lumaWidth >> sh, *h = scs->lumaHeight >> sh;
    *stride = plane == 0 ? pic->strideY : plane == 1 ? pic->strideCb : pic->strideCr;
    *origin = (size_t)(pic->originX >> sh) + (size_t)(pic->originY >> sh) * *stride;
}

static uint8_t *plane_base(const EbPictureBufferDesc_t *pic, int plane)
{
    return plane == 0 ? pic->bufferY : plane == 1 ? pic->bufferCb : pic->bufferCr;
}

/* must hold g_lock */
static DlfPicture *find_picture(PictureControlSet_t *pcs, const EbPictureBufferDesc_t *pic, uint32_t bps)
{
    DlfPicture *slot = NULL;
    for (int i = 0; i < MAX_INFLIGHT; i++) {
        if (g_pics[i].pcs == pcs && g_pics[i].picture_number == pcs->pictureNumber)
            return &g_pics[i];
        if (!g_pics[i].pcs && !slot)
            slot = &g_pics[i];
    }
    if (!slot)
        return NULL;
    const SequenceControlSet_t *scs = (const SequenceControlSet_t *)pcs->sequenceControlSetWrapperPtr->objectPtr;
    memset(slot, 0, sizeof(*slot));
    slot->pcs = pcs, slot->picture_number = pcs->pictureNumber, slot->bps = bps;
    slot->nlcu = ((scs->lumaWidth + 63) >> 6) * ((scs->lumaHeight + 63) >> 6);
    for (int p = 0; p < 3; p++) {
        uint32_t w, h, st;
        size_t org;
        plane_geometry(pic, scs, p, &w, &h, &st, &org);
        slot->pre[p] = (uint8_t *)calloc((size_t)w * h, bps);
    }
    slot->bsv = (uint8_t *)calloc(slot->nlcu, 256), slot->bsh = (uint8_t *)calloc(slot->nlcu, 256);
    return slot;
}

static void first_call(EbPictureBufferDesc_t *pic, EB_U32 x0, EB_U32 y0, EB_U32 lw, EB_U32 lh, EB_U8 *bsv, EB_U8 *bsh,
                       PictureControlSet_t *pcs, uint32_t bps)
{
    if (!dump_on() || pic->colorFormat != EB_YUV420)
        return;
    const SequenceControlSet_t *scs = (const SequenceControlSet_t *)pcs->sequenceControlSetWrapperPtr->objectPtr;
    pthread_mutex_lock(&g_lock);
    DlfPicture *d = find_picture(pcs, pic, bps);
    pthread_mutex_unlock(&g_lock);
    if (!d)
        return;
    for (int p = 0; p < 3; p++) {
        uint32_t w, h, st;
        size_t org;
        plane_geometry(pic, scs, p, &w, &h, &st, &org);
        const uint32_t sh = p ? 1 : 0, bx = x0 >> sh, by = y0 >> sh, bw = lw >> sh, bh = lh >> sh;
        const uint8_t *src = plane_base(pic, p);
        for (uint32_t y = 0; y < bh; y++)
            memcpy(d->pre[p] + ((size_t)(by + y) * w + bx) * bps, src + (org + (size_t)(by + y) * st + bx) * bps,
                   (size_t)bw * bps);
    }
    const uint32_t lcu = (y0 >> 6) * ((scs->lumaWidth + 63) >> 6) + (x0 >> 6);
    memcpy(d->bsv + (size_t)lcu * 256, bsv, 256);
    memcpy(d->bsh + (size_t)lcu * 256, bsh, 256);
    pthread_mutex_lock(&g_lock);
    d->seen_first++;
    pthread_mutex_unlock(&g_lock);
}

static void last_call(EbPictureBufferDesc_t *pic, PictureControlSet_t *pcs)
{
    if (!dump_on() || pic->colorFormat != EB_YUV420)
        return;
    const SequenceControlSet_t *scs = (const SequenceControlSet_t *)pcs->sequenceControlSetWrapperPtr->objectPtr;
    pthread_mutex_lock(&g_lock);
    DlfPicture *d = NULL;
    for (int i = 0; i < MAX_INFLIGHT; i++)
        if (g_pics[i].pcs == pcs && g_pics[i].picture_number == pcs->pictureNumber)
            d = &g_pics[i];
    if (!d || ++d->seen_last < d->nlcu) {
        pthread_mutex_unlock(&g_lock);
        return;
    }
    /* every LCU of the picture has finished its three driver calls: write the record */
    DlfRecordHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = DLF_DUMP_MAGIC, h.header_size = (uint32_t)sizeof(h), h.picture_number = pcs->pictureNumber;
    h.width = scs->lumaWidth, h.height = scs->lumaHeight, h.bytes_per_sample = d->bps, h.slice_type = pcs->sliceType;
    h.lcu_cols = (scs->lumaWidth + 63) >> 6, h.lcu_rows = (scs->lumaHeight + 63) >> 6;
    h.qp_stride = pcs->qpArrayStride, h.qp_size = pcs->qpArraySize;
    h.tc_offset = pcs->tcOffset, h.beta_offset = pcs->betaOffset, h.cb_qp_offset = pcs->cbQpOffset, h.cr_qp_offset = pcs->crQpOffset;
    if (d->seen_first == d->nlcu) {
        fwrite(&h, sizeof(h), 1, g_file);
        for (int p = 0; p < 3; p++) {
            uint32_t w, hh, st;
            size_t org;
            plane_geometry(pic, scs, p, &w, &hh, &st, &org);
            fwrite(d->pre[p], d->bps, (size_t)w * hh, g_file);
        }
        for (int p = 0; p < 3; p++) {
            uint32_t w, hh, st;
            size_t org;
            plane_geometry(pic, scs, p, &w, &hh, &st, &org);
            const uint8_t *src = plane_base(pic, p);
            for (uint32_t y = 0; y < hh; y++)
                fwrite(src + (org + (size_t)y * st) * d->bps, d->bps, w, g_file);
        }
        fwrite(d->bsv, 256, d->nlcu, g_file);
        fwrite(d->bsh, 256, d->nlcu, g_file);
        fwrite(pcs->qpArray, 1, pcs->qpArraySize, g_file);
        /* inputs of the boundary-strength derivation (SetBSArrayBasedOnPUBoundary / TUBoundary, :339-530), as picture-level
         * maps: one entry per 8x8 block from the final coding-unit tree, the luma cbf map per 4x4 block, the two reference
         * POCs, the per-LCU tile-edge flags */
        {
            typedef struct CuMapEntry { uint8_t mode, dir, size_log2, pad; int16_t mv[2][2]; } CuMapEntry;
            const uint32_t bw = scs->lumaWidth >> 3, bh = scs->lumaHeight >> 3;
            CuMapEntry *map = (CuMapEntry *)calloc((size_t)bw * bh, sizeof(CuMapEntry));
            uint8_t *edge = (uint8_t *)calloc(d->nlcu, 1);
            for (uint32_t l = 0; l < d->nlcu; l++) {
                LargestCodingUnit_t *lcu = pcs->lcuPtrArray[l];
                edge[l] = (uint8_t)((lcu->lcuEdgeInfoPtr->tileLeftEdgeFlag ? 1 : 0) | (lcu->lcuEdgeInfoPtr->tileTopEdgeFlag ? 2 : 0));
                uint32_t leaf = 0;
                while (leaf < CU_MAX_COUNT) {
                    const CodingUnit_t *cu = lcu->codedLeafArrayPtr[leaf];
                    const CodedUnitStats_t *st = GetCodedUnitStats(leaf);
                    if (cu->splitFlag == EB_FALSE) {
                        const uint32_t x0 = lcu->originX + st->originX, y0 = lcu->originY + st->originY;
                        for (uint32_t y = y0; y < y0 + st->size && y < scs->lumaHeight; y += 8)
                            for (uint32_t x = x0; x < x0 + st->size && x < scs->lumaWidth; x += 8) {
                                CuMapEntry *e = &map[(y >> 3) * bw + (x >> 3)];
                                e->mode = (uint8_t)cu->predictionModeFlag, e->dir = (uint8_t)cu->predictionUnitArray[0].interPredDirectionIndex;
                                e->size_log2 = (uint8_t)st->sizeLog2;
                                e->mv[0][0] = cu->predictionUnitArray[0].mv[0].x, e->mv[0][1] = cu->predictionUnitArray[0].mv[0].y;
                                e->mv[1][0] = cu->predictionUnitArray[0].mv[1].x, e->mv[1][1] = cu->predictionUnitArray[0].mv[1].y;
                            }
                        leaf += DepthOffset[st->depth];
                    } else {
                        leaf++;
                    }
                }
            }
            uint64_t poc[2] = {0, 0};
            for (int l = 0; l < 2; l++)
                if (pcs->sliceType != EB_I_PICTURE && pcs->refPicPtrArray[l])
                    poc[l] = ((EbReferenceObject_t *)pcs->refPicPtrArray[l]->objectPtr)->refPOC;
            const uint32_t tag[4] = {0x31585342U /* "BSX1" */, bw * bh, (scs->lumaWidth >> 2) * (scs->lumaHeight >> 2), d->nlcu};
            fwrite(tag, sizeof(tag), 1, g_file);
            fwrite(map, sizeof(CuMapEntry), (size_t)bw * bh, g_file);
            fwrite(pcs->cbfMapArray, 1, tag[2], g_file);
            fwrite(poc, sizeof(poc), 1, g_file);
            fwrite(edge, 1, d->nlcu, g_file);
            free(map), free(edge);
        }
        fflush(g_file);
    }
    for (int p = 0; p < 3; p++)
        free(d->pre[p]);
    free(d->bsv), free(d->bsh);
    memset(d, 0, sizeof(*d));
    pthread_mutex_unlock(&g_lock);
}

EB_ERRORTYPE __real_LCUInternalAreaDLFCore(EbPictureBufferDesc_t *, EB_U32, EB_U32, EB_U32, EB_U32, EB_U8 *, EB_U8 *, PictureControlSet_t *);
EB_ERRORTYPE __real_LCUInternalAreaDLFCore16bit(EbPictureBufferDesc_t *, EB_U32, EB_U32, EB_U32, EB_U32, EB_U8 *, EB_U8 *, PictureControlSet_t *);
void __real_LCUPictureEdgeDLFCore(EbPictureBufferDesc_t *, EB_U32, EB_U32, EB_U32, EB_U32, EB_U32, PictureControlSet_t *);
void __real_LCUPictureEdgeDLFCore16bit(EbPictureBufferDesc_t *, EB_U32, EB_U32, EB_U32, EB_U32, EB_U32, PictureControlSet_t *);

EB_ERRORTYPE __wrap_LCUInternalAreaDLFCore(EbPictureBufferDesc_t *pic, EB_U32 x, EB_U32 y, EB_U32 w, EB_U32 h, EB_U8 *bsv,
                                           EB_U8 *bsh, PictureControlSet_t *pcs)
{
    first_call(pic, x, y, w, h, bsv, bsh, pcs, 1);
    return __real_LCUInternalAreaDLFCore(pic, x, y, w, h, bsv, bsh, pcs);
}
EB_ERRORTYPE __wrap_LCUInternalAreaDLFCore16bit(EbPictureBufferDesc_t *pic, EB_U32 x, EB_U32 y, EB_U32 w, EB_U32 h, EB_U8 *bsv,
                                                EB_U8 *bsh, PictureControlSet_t *pcs)
{
    first_call(pic, x, y, w, h, bsv, bsh, pcs, 2);
    return __real_LCUInternalAreaDLFCore16bit(pic, x, y, w, h, bsv, bsh, pcs);
}
void __wrap_LCUPictureEdgeDLFCore(EbPictureBufferDesc_t *pic, EB_U32 idx, EB_U32 x, EB_U32 y, EB_U32 w, EB_U32 h,
                                  PictureControlSet_t *pcs)
{
    __real_LCUPictureEdgeDLFCore(pic, idx, x, y, w, h, pcs);
    last_call(pic, pcs);
}
void __wrap_LCUPictureEdgeDLFCore16bit(EbPictureBufferDesc_t *pic, EB_U32 idx, EB_U32 x, EB_U32 y, EB_U32 w, EB_U32 h,
                                       PictureControlSet_t *pcs)
{
    __real_LCUPictureEdgeDLFCore16bit(pic, idx, x, y, w, h, pcs);
    last_call(pic, pcs);
}
