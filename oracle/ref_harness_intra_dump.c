/*
 * TEST INFRASTRUCTURE ONLY (oracle/): interposer on the REFERENCE's encode-pass intra prediction of a prediction unit,
 * the pair GenerateIntraReferenceSamplesEncodePass (Codec/EbIntraPrediction.c:212; 16-bit twin
 * GenerateIntraReference16bitSamplesEncodePass) + EncodePassIntraPrediction (:4395; 16-bit :4680), which the encode pass
 * reaches through the global tables GenerateIntraReferenceSamplesFuncTable[2] / EncodePassIntraPredictionFuncTable[2]
 * (Codec/EbCodingLoop.c:1814, :1832).  A constructor swaps the slots for recording wrappers.  Compiled only into
 * oracle/_ref/libsvtref.so.
 *
 * With SVT_REF_INTRA_DUMP=<file>, a sample of the call pairs (every SVT_REF_INTRA_STRIDE-th, default 7) leaves one binary
 * record: the slices of the neighbour arrays the first call may look at (mode type per 4 samples, reconstructed luma /
 * chroma samples left, above and above-left of the unit), its flags, the two z-order availabilities it derives, and the
 * modes and the three predicted blocks of the second call.  tests/golden/make_intra_golden.py builds the fixtures.
 *
 * The mode decision's closed-loop intra prediction IntraPredictionCl (Codec/EbIntraPrediction.c:3682, reached through
 * ProductPredictionFunTableCl, EbProductCodingLoop.c:223-227) is caught with -Wl,--wrap=IntraPredictionCl: with
 * SVT_REF_INTRA_MD_DUMP=<file> every SVT_REF_INTRA_MD_STRIDE-th call (default 23) leaves records of the same layout, one for
 * the luma block (component_mask 1, the tile-edge flags GenerateIntraLumaReferenceSamplesMd derives, EbProductCodingLoop.c:
 * 274-276) and / or one for the two chroma blocks (component_mask 6, no edge flags, :2203-2219), cut from the mode decision's
 * own neighbour arrays and its candidate prediction buffer.  Its open-loop twin IntraPredictionOl (:5427,
 * -Wl,--wrap=IntraPredictionOl) leaves records of the same layout too, with the neighbours cut from the source picture as
 * UpdateNeighborSamplesArrayOL / UpdateChromaNeighborSamplesArrayOL do (:4952, :5065), every group marked available and
 * pad0 = 1 ("no smoothing").
 * No reference source here.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "EbDefinitions.h"
#include "EbPictureBufferDesc.h"
#include "EbNeighborArrays.h"
#include "EbIntraPrediction.h"
#include "EbAvailability.h"
#include "EbPictureControlSet.h"
#include "EbModeDecisionProcess.h"
#include "EbModeDecision.h"

typedef EB_ERRORTYPE (*GenFn)(EB_BOOL, EB_BOOL, EB_U32, EB_U32, EB_U32, EB_U32, EB_U32, NeighborArrayUnit_t *, NeighborArrayUnit_t *,
                              NeighborArrayUnit_t *, NeighborArrayUnit_t *, void *, EB_COLOR_FORMAT, EB_BOOL, EB_BOOL, EB_BOOL);
typedef EB_ERRORTYPE (*PredFn)(void *, EB_U32, EB_U32, EB_U32, EB_U32, EbPictureBufferDesc_t *, EB_COLOR_FORMAT, EB_BOOL, EB_U32, EB_U32,
                               EB_U32);
typedef EB_ERRORTYPE (*LumaGenFn)(EB_BOOL, EB_BOOL, EB_U32, EB_U32, EB_U32, EB_U32, EB_U32, NeighborArrayUnit_t *, NeighborArrayUnit_t *,
                                  NeighborArrayUnit_t *, NeighborArrayUnit_t *, void *, EB_BOOL, EB_BOOL, EB_BOOL);
typedef EB_ERRORTYPE (*ChromaGenFn)(EB_BOOL, EB_BOOL, EB_U32, EB_U32, EB_U32, EB_U32, EB_U32, NeighborArrayUnit_t *, NeighborArrayUnit_t *,
                                    NeighborArrayUnit_t *, NeighborArrayUnit_t *, void *, EB_COLOR_FORMAT, EB_BOOL, EB_BOOL, EB_BOOL, EB_BOOL);
/* the intra 4x4 path of the encode pass (EbCodingLoop.c:3594-3690): luma per 4x4 partition, chroma once per 8x8 coding unit */
extern LumaGenFn GenerateLumaIntraReferenceSamplesFuncTable[2];
extern ChromaGenFn GenerateChromaIntraReferenceSamplesFuncTable[2];
static LumaGenFn g_lgen[2];
static ChromaGenFn g_cgen[2];
extern GenFn GenerateIntraReferenceSamplesFuncTable[2];
extern PredFn EncodePassIntraPredictionFuncTable[2];
static GenFn g_gen[2];
static PredFn g_pred[2];

#define INTRA_DUMP_MAGIC 0x52544e49U /* "INTR" */
typedef struct IntraRecord {
    uint32_t magic, record_size;
    uint32_t size, bytes_per_sample;
    uint8_t constrained_intra, strong_smoothing, pic_left, pic_top, pic_right, bottom_left_ok, top_right_ok, pad0;
    uint32_t luma_mode, chroma_mode, component_mask, pad1;
    uint8_t mode_left[32], mode_top[32], mode_tl, pad2[7]; /* per 4 samples going down / right from the unit's corner; 0xFE: beyond the array */
    uint16_t left[3][128], top[3][128], tl[3], pad3;       /* [plane][i]: i-th sample below the corner / right of it */
    uint16_t pred_y[64 * 64], pred_cb[32 * 32], pred_cr[32 * 32]; /* size x size (chroma size/2), row pitch = that size */
} IntraRecord;

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static FILE *g_file;
static int g_state, g_stride = 7;
static unsigned long g_calls;
static __thread IntraRecord *t_pending;
static __thread void *t_pending_ref;
static __thread IntraRecord *t_pending4[2]; /* intra 4x4: [0] a 4x4 luma partition, [1] the chroma pair of its 8x8 coding unit */
static __thread void *t_pending4_ref[2];
static FILE *g_file4;                        /* SVT_REF_INTRA4_DUMP, every SVT_REF_INTRA4_STRIDE-th generator call (default 5) */
static int g_state4, g_stride4 = 5;
static unsigned long g_calls4;

static uint16_t rd(const uint8_t *a, uint32_t i, int bps) { return bps == 1 ? a[i] : ((const uint16_t *)a)[i]; }

static EB_ERRORTYPE gen_wrapper(int is16, EB_BOOL constrained, EB_BOOL strong, EB_U32 originX, EB_U32 originY, EB_U32 size, EB_U32 lcuSize,
                                EB_U32 cuDepth, NeighborArrayUnit_t *mode, NeighborArrayUnit_t *y, NeighborArrayUnit_t *cb,
                                NeighborArrayUnit_t *cr, void *ref, EB_COLOR_FORMAT cf, EB_BOOL pl, EB_BOOL pt, EB_BOOL pr)
{
    if (g_state == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_state == 0) {
            const char *path = getenv("SVT_REF_INTRA_DUMP"), *st = getenv("SVT_REF_INTRA_STRIDE");
            g_file = path ? fopen(path, "wb") : NULL;
            if (st && atoi(st) > 0)
                g_stride = atoi(st);
            g_state = g_file ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
    free(t_pending);
    t_pending = NULL;
    int take = 0;
    if (g_state > 0 && cf == EB_YUV420 && size >= 8 && size <= 64) {
        pthread_mutex_lock(&g_lock);
        take = (g_calls++ % (unsigned long)g_stride) == 0;
        pthread_mutex_unlock(&g_lock);
    }
    if (take) {
        IntraRecord *r = (IntraRecord *)calloc(1, sizeof(*r));
        const int bps = is16 ? 2 : 1;
        r->magic = INTRA_DUMP_MAGIC, r->record_size = (uint32_t)sizeof(*r), r->size = size, r->bytes_per_sample = (uint32_t)bps;
        r->constrained_intra = constrained, r->strong_smoothing = strong, r->pic_left = pl, r->pic_top = pt, r->pic_right = pr;
        uint32_t lg = 0;
        while ((1u << lg) < size)
            lg++;
        const uint32_t cuIndex = ((originY & (lcuSize - 1)) >> lg) * (1u << cuDepth) + ((originX & (lcuSize - 1)) >> lg);
        r->bottom_left_ok = isBottomLeftAvailable(cuDepth, cuIndex), r->top_right_ok = isUpperRightAvailable(cuDepth, cuIndex);
        for (uint32_t k = 0; k < 2 * size / 4; k++) {
            const uint32_t li = GetNeighborArrayUnitLeftIndex(mode, originY + 4 * k), ti = GetNeighborArrayUnitTopIndex(mode, originX + 4 * k);
            r->mode_left[k] = li >= mode->leftArraySize ? 0xFE : mode->leftArray[li];
            r->mode_top[k] = ti >= mode->topArraySize ? 0xFE : mode->topArray[ti];
        }
        r->mode_tl = mode->topLeftArray[GetNeighborArrayUnitTopLeftIndex(mode, (EB_S32)originX, (EB_S32)originY)];
        NeighborArrayUnit_t *na[3] = {y, cb, cr};
        for (int p = 0; p < 3; p++) {
            const uint32_t sh = p ? 1 : 0, n2 = (2 * size) >> sh, ox = originX >> sh, oy = originY >> sh;
            for (uint32_t i = 0; i < n2; i++) {
                /* a sample is only read when its 4-sample mode entry lies inside the mode array */
                const uint32_t k = (i << sh) >> 2;
                r->left[p][i] = r->mode_left[k] == 0xFE ? 0 : rd(na[p]->leftArray, oy + i, bps);
                r->top[p][i] = r->mode_top[k] == 0xFE ? 0 : rd(na[p]->topArray, ox + i, bps);
            }
            r->tl[p] = p == 0 ? rd(y->topLeftArray, MAX_PICTURE_HEIGHT_SIZE + originX - originY, bps)
                              : rd(na[p]->topLeftArray, ((MAX_PICTURE_HEIGHT_SIZE - originY) >> 1) + (originX >> 1), bps);
        }
        t_pending = r, t_pending_ref = ref;
    }
    return g_gen[is16](constrained, strong, originX, originY, size, lcuSize, cuDepth, mode, y, cb, cr, ref, cf, pl, pt, pr);
}

/* neighbour-array slices of a unit of `size` at (originX, originY) for the planes [p0, p1) */
static IntraRecord *slices_record(int is16, EB_BOOL constrained, EB_BOOL strong, EB_U32 originX, EB_U32 originY, EB_U32 size, EB_U32 lcuSize,
                                  EB_U32 partitionDepth, NeighborArrayUnit_t *mode, NeighborArrayUnit_t *na[3], int p0, int p1, EB_BOOL pl,
                                  EB_BOOL pt, EB_BOOL pr)
{
    IntraRecord *r = (IntraRecord *)calloc(1, sizeof(*r));
    const int bps = is16 ? 2 : 1;
    r->magic = INTRA_DUMP_MAGIC, r->record_size = (uint32_t)sizeof(*r), r->size = size, r->bytes_per_sample = (uint32_t)bps;
    r->constrained_intra = constrained, r->strong_smoothing = strong, r->pic_left = pl, r->pic_top = pt, r->pic_right = pr;
    uint32_t lg = 0;
    while ((1u << lg) < size)
        lg++;
    const uint32_t cuIndex = ((originY & (lcuSize - 1)) >> lg) * (1 << partitionDepth) + ((originX & (lcuSize - 1)) >> lg);
    r->bottom_left_ok = isBottomLeftAvailable(partitionDepth, cuIndex), r->top_right_ok = isUpperRightAvailable(partitionDepth, cuIndex);
    for (uint32_t k = 0; k < 2 * size / 4; k++) {
        const uint32_t li = GetNeighborArrayUnitLeftIndex(mode, originY + 4 * k), ti = GetNeighborArrayUnitTopIndex(mode, originX + 4 * k);
        r->mode_left[k] = li >= mode->leftArraySize ? 0xFE : mode->leftArray[li];
        r->mode_top[k] = ti >= mode->topArraySize ? 0xFE : mode->topArray[ti];
    }
    r->mode_tl = mode->topLeftArray[GetNeighborArrayUnitTopLeftIndex(mode, (EB_S32)originX, (EB_S32)originY)];
    for (int p = p0; p < p1; p++) {
        const uint32_t sh = p ? 1 : 0, n2 = (2 * size) >> sh, ox = originX >> sh, oy = originY >> sh;
        for (uint32_t i = 0; i < n2; i++) {
            const uint32_t k = (i << sh) >> 2;
            r->left[p][i] = r->mode_left[k] == 0xFE ? 0 : rd(na[p]->leftArray, oy + i, bps);
            r->top[p][i] = r->mode_top[k] == 0xFE ? 0 : rd(na[p]->topArray, ox + i, bps);
        }
        r->tl[p] = p == 0 ? rd(na[0]->topLeftArray, MAX_PICTURE_HEIGHT_SIZE + originX - originY, bps)
                          : rd(na[p]->topLeftArray, ((MAX_PICTURE_HEIGHT_SIZE - originY) >> 1) + (originX >> 1), bps);
    }
    return r;
}

static int take4(void)
{
    if (g_state4 == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_state4 == 0) {
            const char *path = getenv("SVT_REF_INTRA4_DUMP"), *st = getenv("SVT_REF_INTRA4_STRIDE");
            g_file4 = path ? fopen(path, "wb") : NULL;
            if (st && atoi(st) > 0)
                g_stride4 = atoi(st);
            g_state4 = g_file4 ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
    if (g_state4 < 0)
        return 0;
    pthread_mutex_lock(&g_lock);
    const int take = (g_calls4++ % (unsigned long)g_stride4) == 0;
    pthread_mutex_unlock(&g_lock);
    return take;
}

static EB_ERRORTYPE lgen_wrapper(int is16, EB_BOOL constrained, EB_BOOL strong, EB_U32 originX, EB_U32 originY, EB_U32 size, EB_U32 lcuSize,
                                 EB_U32 cuDepth, NeighborArrayUnit_t *mode, NeighborArrayUnit_t *y, NeighborArrayUnit_t *cb, NeighborArrayUnit_t *cr,
                                 void *ref, EB_BOOL pl, EB_BOOL pt, EB_BOOL pr)
{
    free(t_pending4[0]);
    t_pending4[0] = NULL;
    if (size == 4 && !getenv("SVT_REF_INTRA4_MD") && take4()) {
        NeighborArrayUnit_t *na[3] = {y, cb, cr};
        t_pending4[0] = slices_record(is16, constrained, strong, originX, originY, 4, lcuSize, cuDepth + 1, mode, na, 0, 1, pl, pt, pr);
        t_pending4_ref[0] = ref;
    }
    return g_lgen[is16](constrained, strong, originX, originY, size, lcuSize, cuDepth, mode, y, cb, cr, ref, pl, pt, pr);
}

static EB_ERRORTYPE cgen_wrapper(int is16, EB_BOOL constrained, EB_BOOL strong, EB_U32 originX, EB_U32 originY, EB_U32 size, EB_U32 lcuSize,
                                 EB_U32 cuDepth, NeighborArrayUnit_t *mode, NeighborArrayUnit_t *y, NeighborArrayUnit_t *cb, NeighborArrayUnit_t *cr,
                                 void *ref, EB_COLOR_FORMAT cf, EB_BOOL second, EB_BOOL pl, EB_BOOL pt, EB_BOOL pr)
{
    free(t_pending4[1]);
    t_pending4[1] = NULL;
    if (size == 8 && cf == EB_YUV420 && !second && !getenv("SVT_REF_INTRA4_MD") && take4()) {
        NeighborArrayUnit_t *na[3] = {y, cb, cr};
        t_pending4[1] = slices_record(is16, constrained, strong, originX, originY, 8, lcuSize, cuDepth, mode, na, 1, 3, pl, pt, pr);
        t_pending4_ref[1] = ref;
    }
    return g_cgen[is16](constrained, strong, originX, originY, size, lcuSize, cuDepth, mode, y, cb, cr, ref, cf, second, pl, pt, pr);
}

static EB_ERRORTYPE pred_wrapper(int is16, void *ref, EB_U32 originX, EB_U32 originY, EB_U32 puSize, EB_U32 puChromaSize,
                                 EbPictureBufferDesc_t *pic, EB_COLOR_FORMAT cf, EB_BOOL second, EB_U32 lumaMode, EB_U32 chromaMode,
                                 EB_U32 mask)
{
    const EB_ERRORTYPE rc = g_pred[is16](ref, originX, originY, puSize, puChromaSize, pic, cf, second, lumaMode, chromaMode, mask);
    if (puSize == 4 && cf == EB_YUV420 && (mask == PICTURE_BUFFER_DESC_LUMA_MASK || mask == PICTURE_BUFFER_DESC_CHROMA_MASK)) {
        const int c = mask == PICTURE_BUFFER_DESC_CHROMA_MASK;
        IntraRecord *q = t_pending4[c];
        t_pending4[c] = NULL;
        if (q && t_pending4_ref[c] == ref && !second) {
            const int bps = (int)q->bytes_per_sample;
            q->luma_mode = lumaMode, q->chroma_mode = chromaMode, q->component_mask = mask;
            for (uint32_t yy = 0; yy < 4; yy++)
                for (uint32_t xx = 0; xx < 4; xx++) {
                    if (!c) {
                        q->pred_y[yy * 4 + xx] = rd(pic->bufferY, (originY + yy) * pic->strideY + originX + xx, bps);
                    } else {
                        q->pred_cb[yy * 4 + xx] = rd(pic->bufferCb, ((originY >> 1) + yy) * pic->strideCb + (originX >> 1) + xx, bps);
                        q->pred_cr[yy * 4 + xx] = rd(pic->bufferCr, ((originY >> 1) + yy) * pic->strideCr + (originX >> 1) + xx, bps);
                    }
                }
            pthread_mutex_lock(&g_lock);
            fwrite(q, sizeof(*q), 1, g_file4);
            fflush(g_file4);
            pthread_mutex_unlock(&g_lock);
        }
        free(q);
        return rc;
    }
    IntraRecord *r = t_pending;
    t_pending = NULL;
    if (!r)
        return rc;
    if (ref != t_pending_ref || puSize != r->size || second || mask != PICTURE_BUFFER_DESC_FULL_MASK) {
        free(r);
        return rc;
    }
    const int bps = (int)r->bytes_per_sample;
    r->luma_mode = lumaMode, r->chroma_mode = chromaMode, r->component_mask = mask;
    for (uint32_t yy = 0; yy < puSize; yy++)
        for (uint32_t xx = 0; xx < puSize; xx++)
            r->pred_y[yy * puSize + xx] = rd(pic->bufferY, (originY + yy) * pic->strideY + originX + xx, bps);
    const uint32_t c = puSize >> 1;
    for (uint32_t yy = 0; yy < c; yy++)
        for (uint32_t xx = 0; xx < c; xx++) {
            r->pred_cb[yy * c + xx] = rd(pic->bufferCb, ((originY >> 1) + yy) * pic->strideCb + (originX >> 1) + xx, bps);
            r->pred_cr[yy * c + xx] = rd(pic->bufferCr, ((originY >> 1) + yy) * pic->strideCr + (originX >> 1) + xx, bps);
        }
    pthread_mutex_lock(&g_lock);
    fwrite(r, sizeof(*r), 1, g_file);
    fflush(g_file);
    pthread_mutex_unlock(&g_lock);
    free(r);
    return rc;
}

#define GEN_ARGS EB_BOOL a, EB_BOOL b, EB_U32 c, EB_U32 d, EB_U32 e, EB_U32 f, EB_U32 g, NeighborArrayUnit_t *h, NeighborArrayUnit_t *i, \
                 NeighborArrayUnit_t *j, NeighborArrayUnit_t *k, void *l, EB_COLOR_FORMAT m, EB_BOOL n, EB_BOOL o, EB_BOOL p
static EB_ERRORTYPE gen8(GEN_ARGS) { return gen_wrapper(0, a, b, c, d, e, f, g, h, i, j, k, l, m, n, o, p); }
static EB_ERRORTYPE gen16(GEN_ARGS) { return gen_wrapper(1, a, b, c, d, e, f, g, h, i, j, k, l, m, n, o, p); }
#define PRED_ARGS void *a, EB_U32 b, EB_U32 c, EB_U32 d, EB_U32 e, EbPictureBufferDesc_t *f, EB_COLOR_FORMAT g, EB_BOOL h, EB_U32 i, EB_U32 j, EB_U32 k
static EB_ERRORTYPE pred8(PRED_ARGS) { return pred_wrapper(0, a, b, c, d, e, f, g, h, i, j, k); }
static EB_ERRORTYPE pred16(PRED_ARGS) { return pred_wrapper(1, a, b, c, d, e, f, g, h, i, j, k); }

#define LGEN_ARGS EB_BOOL a, EB_BOOL b, EB_U32 c, EB_U32 d, EB_U32 e, EB_U32 f, EB_U32 g, NeighborArrayUnit_t *h, NeighborArrayUnit_t *i, \
                  NeighborArrayUnit_t *j, NeighborArrayUnit_t *k, void *l, EB_BOOL n, EB_BOOL o, EB_BOOL p
static EB_ERRORTYPE lgen8(LGEN_ARGS) { return lgen_wrapper(0, a, b, c, d, e, f, g, h, i, j, k, l, n, o, p); }
static EB_ERRORTYPE lgen16(LGEN_ARGS) { return lgen_wrapper(1, a, b, c, d, e, f, g, h, i, j, k, l, n, o, p); }
#define CGEN_ARGS EB_BOOL a, EB_BOOL b, EB_U32 c, EB_U32 d, EB_U32 e, EB_U32 f, EB_U32 g, NeighborArrayUnit_t *h, NeighborArrayUnit_t *i, \
                  NeighborArrayUnit_t *j, NeighborArrayUnit_t *k, void *l, EB_COLOR_FORMAT m, EB_BOOL q, EB_BOOL n, EB_BOOL o, EB_BOOL p
static EB_ERRORTYPE cgen8(CGEN_ARGS) { return cgen_wrapper(0, a, b, c, d, e, f, g, h, i, j, k, l, m, q, n, o, p); }
static EB_ERRORTYPE cgen16(CGEN_ARGS) { return cgen_wrapper(1, a, b, c, d, e, f, g, h, i, j, k, l, m, q, n, o, p); }

__attribute__((constructor)) static void install(void)
{
    g_gen[0] = GenerateIntraReferenceSamplesFuncTable[0], g_gen[1] = GenerateIntraReferenceSamplesFuncTable[1];
    g_pred[0] = EncodePassIntraPredictionFuncTable[0], g_pred[1] = EncodePassIntraPredictionFuncTable[1];
    GenerateIntraReferenceSamplesFuncTable[0] = gen8, GenerateIntraReferenceSamplesFuncTable[1] = gen16;
    g_lgen[0] = GenerateLumaIntraReferenceSamplesFuncTable[0], g_lgen[1] = GenerateLumaIntraReferenceSamplesFuncTable[1];
    g_cgen[0] = GenerateChromaIntraReferenceSamplesFuncTable[0], g_cgen[1] = GenerateChromaIntraReferenceSamplesFuncTable[1];
    GenerateLumaIntraReferenceSamplesFuncTable[0] = lgen8, GenerateLumaIntraReferenceSamplesFuncTable[1] = lgen16;
    GenerateChromaIntraReferenceSamplesFuncTable[0] = cgen8, GenerateChromaIntraReferenceSamplesFuncTable[1] = cgen16;
    EncodePassIntraPredictionFuncTable[0] = pred8, EncodePassIntraPredictionFuncTable[1] = pred16;
}

/* ---- mode-decision side ---- */
EB_ERRORTYPE __real_IntraPredictionCl(ModeDecisionContext_t *md, EB_U32 componentMask, PictureControlSet_t *pcs,
                                      ModeDecisionCandidateBuffer_t *cand);
void svt_ref_fastloop_note_prediction(const void *candidateBuffer);
static FILE *g_md_file;
static int g_md_state, g_md_stride = 23;
static unsigned long g_md_calls;

static void md_open(void)
{
    if (g_md_state == 0) {
        pthread_mutex_lock(&g_lock);
        if (g_md_state == 0) {
            const char *path = getenv("SVT_REF_INTRA_MD_DUMP"), *st = getenv("SVT_REF_INTRA_MD_STRIDE");
            g_md_file = path ? fopen(path, "wb") : NULL;
            if (st && atoi(st) > 0)
                g_md_stride = atoi(st);
            g_md_state = g_md_file ? 1 : -1;
        }
        pthread_mutex_unlock(&g_lock);
    }
}

/* open loop: the 2N left / 1 / 2N top neighbours of a block of a source plane, mid-grey beyond the picture */
static void ol_slices(uint16_t *left, uint16_t *top, uint16_t *tl, const uint8_t *plane, uint32_t stride, uint32_t x, uint32_t y, uint32_t n,
                      uint32_t width, uint32_t height, int picLeft, int picTop)
{
    const uint8_t *src = plane + (size_t)y * stride + x;
    for (uint32_t i = 0; i < 2 * n; i++)
        left[i] = top[i] = 128;
    *tl = 128;
    if (!picLeft)
        for (uint32_t i = 0; i < 2 * n && y + i < height; i++)
            left[i] = src[(size_t)i * stride - 1];
    if (!picLeft && !picTop)
        *tl = src[-(ptrdiff_t)stride - 1];
    if (!picTop)
        for (uint32_t i = 0; i < 2 * n && x + i < width; i++)
            top[i] = src[i - (ptrdiff_t)stride];
}

static void md_record(ModeDecisionContext_t *md, ModeDecisionCandidateBuffer_t *cand, int chroma)
{
    IntraRecord *r = (IntraRecord *)calloc(1, sizeof(*r));
    const uint32_t size = md->cuStats->size, originX = md->cuOriginX, originY = md->cuOriginY, cuDepth = md->cuStats->depth;
    NeighborArrayUnit_t *mode = md->modeTypeNeighborArray;
    r->magic = INTRA_DUMP_MAGIC, r->record_size = (uint32_t)sizeof(*r), r->size = size, r->bytes_per_sample = 1;
    r->constrained_intra = 0, r->strong_smoothing = 1;
    if (!chroma) {
        const uint32_t m = md->lcuPtr->size - 1;
        r->pic_left = md->lcuPtr->lcuEdgeInfoPtr->tileLeftEdgeFlag == EB_TRUE && (originX & m) == 0;
        r->pic_top = md->lcuPtr->lcuEdgeInfoPtr->tileTopEdgeFlag == EB_TRUE && (originY & m) == 0;
        r->pic_right = md->lcuPtr->lcuEdgeInfoPtr->tileRightEdgeFlag == EB_TRUE && ((originX + size) & m) == 0;
    }
    uint32_t lg = 0;
    while ((1u << lg) < size)
        lg++;
    const uint32_t cuIndex = ((originY & 63) >> lg) * (1u << cuDepth) + ((originX & 63) >> lg);
    r->bottom_left_ok = isBottomLeftAvailable(cuDepth, cuIndex), r->top_right_ok = isUpperRightAvailable(cuDepth, cuIndex);
    for (uint32_t k = 0; k < 2 * size / 4; k++) {
        const uint32_t li = GetNeighborArrayUnitLeftIndex(mode, originY + 4 * k), ti = GetNeighborArrayUnitTopIndex(mode, originX + 4 * k);
        r->mode_left[k] = li >= mode->leftArraySize ? 0xFE : mode->leftArray[li];
        r->mode_top[k] = ti >= mode->topArraySize ? 0xFE : mode->topArray[ti];
    }
    r->mode_tl = mode->topLeftArray[GetNeighborArrayUnitTopLeftIndex(mode, (EB_S32)originX, (EB_S32)originY)];
    NeighborArrayUnit_t *na[3] = {md->lumaReconNeighborArray, md->cbReconNeighborArray, md->crReconNeighborArray};
    for (int p = chroma ? 1 : 0; p < (chroma ? 3 : 1); p++) {
        const uint32_t sh = p ? 1 : 0, n2 = (2 * size) >> sh, ox = originX >> sh, oy = originY >> sh;
        for (uint32_t i = 0; i < n2; i++) {
            const uint32_t k = (i << sh) >> 2;
            r->left[p][i] = r->mode_left[k] == 0xFE ? 0 : rd(na[p]->leftArray, oy + i, 1);
            r->top[p][i] = r->mode_top[k] == 0xFE ? 0 : rd(na[p]->topArray, ox + i, 1);
        }
        r->tl[p] = p == 0 ? rd(na[0]->topLeftArray, MAX_PICTURE_HEIGHT_SIZE + originX - originY, 1)
                          : rd(na[p]->topLeftArray, ((MAX_PICTURE_HEIGHT_SIZE - originY) >> 1) + (originX >> 1), 1);
    }
    r->luma_mode = cand->candidatePtr->intraLumaMode, r->chroma_mode = 4 /* derived: "the chromaMode is always DM" */;
    r->component_mask = chroma ? PICTURE_BUFFER_DESC_CHROMA_MASK : PICTURE_BUFFER_DESC_LUMA_MASK;
    const EbPictureBufferDesc_t *pred = cand->predictionPtr;
    if (!chroma) {
        const uint32_t o = (originY & 63) * 64 + (originX & 63);
        for (uint32_t yy = 0; yy < size; yy++)
            for (uint32_t xx = 0; xx < size; xx++)
                r->pred_y[yy * size + xx] = pred->bufferY[o + yy * pred->strideY + xx];
    } else {
        const uint32_t o = (((originY & 63) * 32) + (originX & 63)) >> 1, c = size >> 1;
        for (uint32_t yy = 0; yy < c; yy++)
            for (uint32_t xx = 0; xx < c; xx++) {
                r->pred_cb[yy * c + xx] = pred->bufferCb[o + yy * pred->strideCb + xx];
                r->pred_cr[yy * c + xx] = pred->bufferCr[o + yy * pred->strideCr + xx];
            }
    }
    pthread_mutex_lock(&g_lock);
    fwrite(r, sizeof(*r), 1, g_md_file);
    fflush(g_md_file);
    pthread_mutex_unlock(&g_lock);
    free(r);
}

EB_ERRORTYPE __wrap_IntraPredictionCl(ModeDecisionContext_t *md, EB_U32 componentMask, PictureControlSet_t *pcs,
                                      ModeDecisionCandidateBuffer_t *cand)
{
    md_open();
    const EB_ERRORTYPE rc = __real_IntraPredictionCl(md, componentMask, pcs, cand);
    svt_ref_fastloop_note_prediction(cand); /* ref_harness_fastloop_dump.c: this candidate buffer was just predicted */
    int take = 0;
    if (g_md_state > 0 && !getenv("SVT_REF_INTRA_MD_OL") && !md->intraMdOpenLoopFlag && md->cuStats->size >= 8 && md->cuStats->size <= 32) {
        pthread_mutex_lock(&g_lock);
        take = (g_md_calls++ % (unsigned long)g_md_stride) == 0;
        pthread_mutex_unlock(&g_lock);
    }
    if (take && (componentMask & PICTURE_BUFFER_DESC_LUMA_MASK))
        md_record(md, cand, 0);
    if (take && (componentMask & PICTURE_BUFFER_DESC_CHROMA_MASK) == PICTURE_BUFFER_DESC_CHROMA_MASK && md->useChromaInformationInFullLoop)
        md_record(md, cand, 1);
    return rc;
}

EB_ERRORTYPE __real_IntraPredictionOl(ModeDecisionContext_t *md, EB_U32 componentMask, PictureControlSet_t *pcs,
                                      ModeDecisionCandidateBuffer_t *cand);
static void ol_record(ModeDecisionContext_t *md, PictureControlSet_t *pcs, ModeDecisionCandidateBuffer_t *cand, int chroma)
{
    IntraRecord *r = (IntraRecord *)calloc(1, sizeof(*r));
    const uint32_t size = md->cuStats->size, originX = md->cuOriginX, originY = md->cuOriginY, m = md->lcuPtr->size - 1;
    const EbPictureBufferDesc_t *in = pcs->ParentPcsPtr->enhancedPicturePtr;
    const int picLeft = md->lcuPtr->lcuEdgeInfoPtr->pictureLeftEdgeFlag == EB_TRUE && (originX & m) == 0;
    const int picTop = md->lcuPtr->lcuEdgeInfoPtr->pictureTopEdgeFlag == EB_TRUE && (originY & m) == 0;
    r->magic = INTRA_DUMP_MAGIC, r->record_size = (uint32_t)sizeof(*r), r->size = size, r->bytes_per_sample = 1;
    r->pad0 = 1; /* no smoothing */
    r->bottom_left_ok = r->top_right_ok = 1;
    memset(r->mode_left, 2, sizeof(r->mode_left)), memset(r->mode_top, 2, sizeof(r->mode_top)), r->mode_tl = 2;
    if (!chroma) {
        ol_slices(r->left[0], r->top[0], &r->tl[0], in->bufferY + (size_t)in->originY * in->strideY + in->originX, in->strideY, originX, originY,
                  size, in->width, in->height, picLeft, picTop);
    } else {
        ol_slices(r->left[1], r->top[1], &r->tl[1], in->bufferCb + (size_t)(in->originY >> 1) * in->strideCb + (in->originX >> 1), in->strideCb,
                  originX >> 1, originY >> 1, size >> 1, in->width >> 1, in->height >> 1, picLeft, picTop);
        ol_slices(r->left[2], r->top[2], &r->tl[2], in->bufferCr + (size_t)(in->originY >> 1) * in->strideCr + (in->originX >> 1), in->strideCr,
                  originX >> 1, originY >> 1, size >> 1, in->width >> 1, in->height >> 1, picLeft, picTop);
    }
    r->luma_mode = cand->candidatePtr->intraLumaMode, r->chroma_mode = 4;
    r->component_mask = chroma ? PICTURE_BUFFER_DESC_CHROMA_MASK : PICTURE_BUFFER_DESC_LUMA_MASK;
    const EbPictureBufferDesc_t *pred = cand->predictionPtr;
    if (!chroma) {
        const uint32_t o = (originY & 63) * 64 + (originX & 63);
        for (uint32_t yy = 0; yy < size; yy++)
            for (uint32_t xx = 0; xx < size; xx++)
                r->pred_y[yy * size + xx] = pred->bufferY[o + yy * pred->strideY + xx];
    } else {
        const uint32_t o = (((originY & 63) * 32) + (originX & 63)) >> 1, c = size >> 1;
        for (uint32_t yy = 0; yy < c; yy++)
            for (uint32_t xx = 0; xx < c; xx++) {
                r->pred_cb[yy * c + xx] = pred->bufferCb[o + yy * pred->strideCb + xx];
                r->pred_cr[yy * c + xx] = pred->bufferCr[o + yy * pred->strideCr + xx];
            }
    }
    pthread_mutex_lock(&g_lock);
    fwrite(r, sizeof(*r), 1, g_md_file);
    fflush(g_md_file);
    pthread_mutex_unlock(&g_lock);
    free(r);
}

EB_ERRORTYPE __wrap_IntraPredictionOl(ModeDecisionContext_t *md, EB_U32 componentMask, PictureControlSet_t *pcs,
                                      ModeDecisionCandidateBuffer_t *cand)
{
    md_open();
    const EB_ERRORTYPE rc = __real_IntraPredictionOl(md, componentMask, pcs, cand);
    svt_ref_fastloop_note_prediction(cand);
    const char *only = getenv("SVT_REF_INTRA_MD_OL"); /* set: record the open-loop calls instead of the closed-loop ones */
    if (!only || g_md_state <= 0 || !md->intraMdOpenLoopFlag || md->cuStats->size < 8 || md->cuStats->size > 32)
        return rc;
    pthread_mutex_lock(&g_lock);
    const int take = (g_md_calls++ % (unsigned long)g_md_stride) == 0;
    pthread_mutex_unlock(&g_lock);
    if (take && (componentMask & PICTURE_BUFFER_DESC_LUMA_MASK))
        ol_record(md, pcs, cand, 0);
    if (take && (componentMask & PICTURE_BUFFER_DESC_CHROMA_MASK) == PICTURE_BUFFER_DESC_CHROMA_MASK)
        ol_record(md, pcs, cand, 1);
    return rc;
}

/* The mode decision's intra 4x4 search (PerformIntra4x4Search, EbProductCodingLoop.c:2642-2960): Intra4x4InitFastLoop (:2543) builds
 * the partition's luma reference (size 4, depth 3, no edge flags) and the coding unit's chroma reference from the mode decision's
 * neighbour arrays, Intra4x4IntraPredictionCl (EbIntraPrediction.c:3993, -Wl,--wrap) predicts every candidate.  Records go to the
 * SVT_REF_INTRA4_DUMP stream with pad1 = 1. */
EB_ERRORTYPE __real_Intra4x4IntraPredictionCl(EB_U32 puIndex, EB_U32 puOriginX, EB_U32 puOriginY, EB_U32 puWidth, EB_U32 puHeight, EB_U32 lcuSize,
                                              EB_U32 componentMask, PictureControlSet_t *pcs, ModeDecisionCandidateBuffer_t *cand, EB_PTR ctx);
EB_ERRORTYPE __wrap_Intra4x4IntraPredictionCl(EB_U32 puIndex, EB_U32 puOriginX, EB_U32 puOriginY, EB_U32 puWidth, EB_U32 puHeight, EB_U32 lcuSize,
                                              EB_U32 componentMask, PictureControlSet_t *pcs, ModeDecisionCandidateBuffer_t *cand, EB_PTR ctx)
{
    const EB_ERRORTYPE rc = __real_Intra4x4IntraPredictionCl(puIndex, puOriginX, puOriginY, puWidth, puHeight, lcuSize, componentMask, pcs, cand, ctx);
    ModeDecisionContext_t *md = (ModeDecisionContext_t *)ctx;
    if (puWidth != 4 || lcuSize != 64 || md->intraMdOpenLoopFlag || !getenv("SVT_REF_INTRA4_MD") || !take4())
        return rc;
    NeighborArrayUnit_t *na[3] = {md->lumaReconNeighborArray, md->cbReconNeighborArray, md->crReconNeighborArray};
    const EbPictureBufferDesc_t *pred = cand->predictionPtr;
    for (int c = 0; c < 2; c++) {
        if (c && (componentMask & PICTURE_BUFFER_DESC_CHROMA_MASK) != PICTURE_BUFFER_DESC_CHROMA_MASK)
            break;
        IntraRecord *q = c ? slices_record(0, EB_FALSE, EB_TRUE, md->cuOriginX, md->cuOriginY, 8, 64, md->cuDepth, md->modeTypeNeighborArray, na, 1, 3,
                                           EB_FALSE, EB_FALSE, EB_FALSE)
                           : slices_record(0, EB_FALSE, EB_TRUE, puOriginX, puOriginY, 4, 64, 3 + 1, md->modeTypeNeighborArray, na, 0, 1, EB_FALSE,
                                           EB_FALSE, EB_FALSE);
        q->luma_mode = cand->candidatePtr->intraLumaMode, q->chroma_mode = 4, q->pad1 = 1;
        q->component_mask = c ? PICTURE_BUFFER_DESC_CHROMA_MASK : PICTURE_BUFFER_DESC_LUMA_MASK;
        const uint32_t oy_ = ((puOriginY & 63) * pred->strideY) + (puOriginX & 63), oc_ = (((puOriginY & 63) * pred->strideCb) + (puOriginX & 63)) >> 1;
        for (uint32_t yy = 0; yy < 4; yy++)
            for (uint32_t xx = 0; xx < 4; xx++) {
                if (!c) {
                    q->pred_y[yy * 4 + xx] = pred->bufferY[oy_ + yy * pred->strideY + xx];
                } else {
                    q->pred_cb[yy * 4 + xx] = pred->bufferCb[oc_ + yy * pred->strideCb + xx];
                    q->pred_cr[yy * 4 + xx] = pred->bufferCr[oc_ + yy * pred->strideCr + xx];
                }
            }
        pthread_mutex_lock(&g_lock);
        fwrite(q, sizeof(*q), 1, g_file4);
        fflush(g_file4);
        pthread_mutex_unlock(&g_lock);
        free(q);
    }
    return rc;
}
