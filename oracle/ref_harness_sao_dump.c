/*
 * TEST INFRASTRUCTURE ONLY (oracle/): link-time interposer around the REFERENCE's EncodeLcuSaoParameters
 * (Codec/EbEntropyCoding.c:6815, called per LCU from EbEntropyCodingProcess.c:160), the one place outside the encode
 * pass where every LCU's final SaoParameters_t passes by.  Compiled only into oracle/_ref/libsvtref.so with
 * -Wl,--wrap=EncodeLcuSaoParameters.
 *
 * With SVT_REF_SAO_DUMP=<file>, every picture whose LCUs all went through leaves one binary record: the slice's two SAO
 * enable flags and, per LCU, the merge flags, type indices, offsets, band positions and the tile-edge flags
 * ApplySaoOffsetsLcu (Codec/EbEncDecProcess.c:215) reads.  Together with the deblocked picture of
 * ref_harness_dlf_dump.c and the encoder's own reconstruction output (-o) this pins picture-level SAO application:
 * tests/golden/make_dlf_golden.py joins the three.  No reference source here.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbCodingUnit.h"
#include "EbEntropyCoding.h"

#define SAO_DUMP_MAGIC 0x204f4153U /* "SAO " */

typedef struct SaoLcuRecord {
    uint8_t merge_left, merge_up, edge_flags /* 1 left, 2 right, 4 top, 8 bottom (tile edges) */, pad;
    uint32_t type[2];
    int32_t offset[3][4];
    uint32_t band[3];
} SaoLcuRecord;
typedef struct SaoRecordHeader {
    uint32_t magic, header_size;
    uint64_t picture_number;
    uint32_t nlcu, sao_flag[2], pad;
} SaoRecordHeader;

typedef struct SaoPicture {
    PictureControlSet_t *pcs;
    uint64_t picture_number;
    uint32_t seen, nlcu;
    SaoLcuRecord *lcus;
} SaoPicture;

#define MAX_INFLIGHT 64
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static SaoPicture g_pics[MAX_INFLIGHT];
static FILE *g_file;
static int g_state;

EB_ERRORTYPE __real_EncodeLcuSaoParameters(LargestCodingUnit_t *tbPtr, EntropyCoder_t *entropyCoderPtr, EB_BOOL saoLumaSliceEnable,
                                           EB_BOOL saoChromaSliceEnable, EB_U8 bitdepth);

EB_ERRORTYPE __wrap_EncodeLcuSaoParameters(LargestCodingUnit_t *tbPtr, EntropyCoder_t *entropyCoderPtr, EB_BOOL saoLumaSliceEnable,
                                           EB_BOOL saoChromaSliceEnable, EB_U8 bitdepth)
{
    EB_ERRORTYPE rc = __real_EncodeLcuSaoParameters(tbPtr, entropyCoderPtr, saoLumaSliceEnable, saoChromaSliceEnable, bitdepth);
    if (g_state == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_state == 0) {
            const char *path = getenv("SVT_REF_SAO_DUMP");
            g_file = path ? fopen(path, "wb") : NULL;
            g_state = g_file ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
    if (g_state < 0)
        return rc;
    PictureControlSet_t *pcs = tbPtr->pictureControlSetPtr;
    pthread_mutex_lock(&g_lock);
    SaoPicture *d = NULL, *slot = NULL;
    for (int i = 0; i < MAX_INFLIGHT; i++) {
        if (g_pics[i].pcs == pcs && g_pics[i].picture_number == pcs->pictureNumber)
            d = &g_pics[i];
        if (!g_pics[i].pcs && !slot)
            slot = &g_pics[i];
    }
    if (!d && slot) {
        d = slot;
        memset(d, 0, sizeof(*d));
        d->pcs = pcs, d->picture_number = pcs->pictureNumber, d->nlcu = pcs->lcuTotalCount;
        d->lcus = (SaoLcuRecord *)calloc(d->nlcu, sizeof(SaoLcuRecord));
    }
    if (d) {
        SaoLcuRecord *r = &d->lcus[tbPtr->index];
        const SaoParameters_t *s = &tbPtr->saoParams;
        r->merge_left = s->saoMergeLeftFlag, r->merge_up = s->saoMergeUpFlag;
        r->edge_flags = (uint8_t)((tbPtr->lcuEdgeInfoPtr->tileLeftEdgeFlag ? 1 : 0) | (tbPtr->lcuEdgeInfoPtr->tileRightEdgeFlag ? 2 : 0) |
                                  (tbPtr->lcuEdgeInfoPtr->tileTopEdgeFlag ? 4 : 0) |
                                  ((tbPtr->tileInfoPtr->tileLcuEndY * MAX_LCU_SIZE <= tbPtr->originY + MAX_LCU_SIZE) ? 8 : 0));
        r->type[0] = s->saoTypeIndex[0], r->type[1] = s->saoTypeIndex[1];
        memcpy(r->offset, s->saoOffset, sizeof(r->offset));
        memcpy(r->band, s->saoBandPosition, sizeof(r->band));
        if (++d->seen == d->nlcu) {
            SaoRecordHeader h;
            memset(&h, 0, sizeof(h));
            h.magic = SAO_DUMP_MAGIC, h.header_size = (uint32_t)sizeof(h), h.picture_number = pcs->pictureNumber, h.nlcu = d->nlcu;
            h.sao_flag[0] = saoLumaSliceEnable, h.sao_flag[1] = saoChromaSliceEnable;
            fwrite(&h, sizeof(h), 1, g_file);
            fwrite(d->lcus, sizeof(SaoLcuRecord), d->nlcu, g_file);
            fflush(g_file);
            free(d->lcus);
            memset(d, 0, sizeof(*d));
        }
    }
    pthread_mutex_unlock(&g_lock);
    return rc;
}
