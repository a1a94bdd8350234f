/*
 * integration/svt_hook_encdec.c - the reference-side binding of the device-resident encode pass (SURVEY 8b `hip_encdec_segment`,
 * include/svt_hevc_amd.h "Device-resident encode pass"), enabled with SVT_HOOK_ENCODEPASS=1.
 *
 * EncodePass (Codec/EbCodingLoop.c:2989) is interposed with -Wl,--wrap=EncodePass.  For an LCU inside what this revision of
 * svt_amd_encode_lcus() / svt_amd_encode_lcus16() covers - 4:2:0, 8-bit or 10-bit (EncodePass with is16bit), every coding unit an
 * intra 2Nx2N unit of 8..32 or an inter 2Nx2N unit of 8..64 (the merge / skip decision of :3838-3882 is made here, from the mode
 * decision's costs, before the call; the reference pictures are device copies kept by svt_hook_me.c), no delta-QP / masking tools,
 * plain quantiser - the binding
 *   1. converts the LCU's final coding-unit tree (LargestCodingUnit_t.codedLeafArrayPtr) and its source samples into the input
 *      contract SvtAmdLcuWork,
 *   2. makes ONE device call: intra reference + prediction, residual, transform, quantiser, inverse transform and reconstruction of
 *      all units and planes of the LCU run on the MI355X against the picture's un-deblocked reconstruction, which stays in HBM,
 *   3. lets the reference's EncodePass run for its bookkeeping (cbf / DC flags, neighbour arrays, boundary strengths, deblocking,
 *      SAO, coefficient buffer for entropy coding) with every compute leaf it reaches answered from the output contract
 *      SvtAmdLcuResult: the intra generator / predictor table slots, EncodePassInterPrediction, PictureResidual / EstimateTransform
 *      return at once, UnifiedQuantizeInvQuantize hands back the device's coefficients, the EncodeGenerateRecon table slot the
 *      device's samples; for AMVP units PictureFullDistortionLuma / TuEstimateCoeffBitsEncDec return at once and EncodeTuCalcCost is
 *      given costs that reproduce the device's luma cbf decision.
 * LCUs outside that (intra 4x4 at encMode <= 2, RDOQ at encMode 0, coefficient shaping at encMode >= 11, ...) are encoded by the
 * reference code; their last
 * row / column and edge mode types (the ep* neighbour arrays after the call) are handed to the device picture before the next
 * device-encoded LCU of the picture needs them.  No fallback on errors: any failure of the HIP library aborts the encoder.
 *
 * Contains no reference source.
 */
#include <pthread.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbEncDecProcess.h"
#include "EbModeDecisionProcess.h"
#include "EbReferenceObject.h"
#include "EbCodingUnit.h"
#include "EbTransformUnit.h"
#include "EbTransforms.h"
#include "EbNeighborArrays.h"
#include "EbUtility.h"
#include "EbAvailability.h"
#include "EbPictureOperators.h"

#include "svt_hook_internal.h"
#include "svt_md_fill.h"

#define EP_LANES 24    /* device contexts (stream + staging) EncDec threads share: one per picture whose device call is in flight */
#define EP_PICTURES 64 /* pictures in flight (PictureControlSet_t objects of the EncDec pool) */

void __real_EncodePass(SequenceControlSet_t *scs, PictureControlSet_t *pcs, LargestCodingUnit_t *lcuPtr, EB_U32 tbAddr, EB_U32 lcuOriginX,
                       EB_U32 lcuOriginY, EB_U32 lcuQp, EB_BOOL enableSaoFlag, EncDecContext_t *contextPtr);
void __real_PictureResidual(EB_U8 *input, EB_U32 inputStride, EB_U8 *pred, EB_U32 predStride, EB_S16 *residual, EB_U32 residualStride,
                            EB_U32 areaWidth, EB_U32 areaHeight);
EB_ERRORTYPE __real_EstimateTransform(EB_S16 *residualBuffer, EB_U32 residualStride, EB_S16 *coeffBuffer, EB_U32 coeffStride, EB_U32 transformSize,
                                      EB_S16 *transformInnerArrayPtr, EB_U32 bitIncrement, EB_BOOL dstTansformFlag,
                                      EB_TRANS_COEFF_SHAPE transCoeffShape);

void AddChromaEncDec(PictureControlSet_t *pictureControlSetPtr, LargestCodingUnit_t *lcuPtr, CodingUnit_t *cuPtr, ModeDecisionContext_t *contextPtr,
                     EncDecContext_t *contextPtrED, EbPictureBufferDesc_t *inputPicturePtr, EB_U32 inputCbOriginIndex, EB_U32 cuChromaOriginIndex,
                     EB_U32 candIdxInput); /* EbProductCodingLoop.c:4158 */
/* svt_hook_me.c: device copies of the picture's reference pictures (uploaded once per reference picture) */
void svt_hook_resident_references(SvtAmdContext *lane, const PictureControlSet_t *pcs, int wide, SvtAmdRefPicture out[2], int have[2], int slot[2]);
void svt_hook_release_references(const int slot[2]);
EB_ERRORTYPE __real_EncodeTuCalcCost(EncDecContext_t *contextPtr, EB_U32 *countNonZeroCoeffs, EB_U64 yTuDistortion[DIST_CALC_TOTAL], EB_U64 *yTuCoeffBits,
                                     EB_U32 componentMask);
EB_ERRORTYPE __real_PictureFullDistortionLuma(EbPictureBufferDesc_t *coeff, EB_U32 coeffLumaOriginIndex, EbPictureBufferDesc_t *reconCoeff,
                                              EB_U32 reconCoeffLumaOriginIndex, EB_U32 areaSize, EB_U64 lumaDistortion[DIST_CALC_TOTAL],
                                              EB_U32 countNonZeroCoeffsY, EB_MODETYPE mode);
EB_ERRORTYPE __real_TuEstimateCoeffBitsEncDec(EB_U32 tuOriginIndex, EB_U32 tuChromaOriginIndex, EntropyCoder_t *entropyCoderPtr,
                                              EbPictureBufferDesc_t *coeffBufferTB, EB_U32 countNonZeroCoeffs[3], EB_U64 *yTuCoeffBits,
                                              EB_U64 *cbTuCoeffBits, EB_U64 *crTuCoeffBits, EB_U32 transformSize, EB_U32 transformChromaSize,
                                              EB_MODETYPE type, CabacCost_t *CabacCost);

typedef struct {
    const PictureControlSet_t *pcs;
    uint64_t picture_plus1;      /* picture the device picture was begun for */
    int tl_lcus;                 /* SVT_HOOK_TIMELINE: LCUs of the current picture through EncodePass so far */
    double tl_first;
    int ref_pins_plus1[2];       /* reference-cache slots (+ 1; 0 = none) pinned for that picture: released when the object moves on to its next picture / is released */
    SvtAmdEncDecPicture *pic;
    pthread_mutex_t lock;        /* pending list + the device put of it */
    void *pending;               /* host-encoded LCUs not handed to the device yet: SvtAmdLcuBorder[] or SvtAmdLcuBorder16[] */
    int npending, cap, wide;
    uint32_t width, height;      /* the picture format the entry's buffers were sized for */
    /* SVT_HOOK_ENCODEPASS_REFS: everything the device needs to finish the picture itself when its last LCU is through */
    void *works_all, *res_all;   /* contract records of every LCU, raster order (8- or 16-bit contract) */
    uint8_t *sao_enable;         /* LCUs the reference ran SaoGenerationDecision for */
    SvtAmdSaoDecisionParams sao_P;
    int sao_any, sao_varies, on_device, done;
    /* SVT_HOOK_MD: the mode decision AND the encode pass of every LCU of the picture in ONE device call, made by the picture's first
     * ModeDecisionLcu call; the later calls (and the EncodePass calls) of the picture are answered from these arrays */
    uint64_t md_picture_plus1;   /* picture the arrays below were filled for */
    uint64_t md_done_plus1;      /* ... and the device call for it has RETURNED (release store): what the lock-free fast path of the per-LCU wraps looks at */
    int md_ok;                   /* 1: served by the device; 0: outside what svt_amd_md_encode_picture covers - the reference code runs */
    SvtAmdMdLcuOut *md_out;      /* pinned host memory (svt_amd_host_alloc), like the staging arrays below: the picture's records move by DMA */
    void *md_works, *md_res;     /* SvtAmdLcuWork / SvtAmdLcuResult of the entry's sample width (wide: the 16-bit twins) */
    uint16_t *md_src16[3];       /* a 10-bit picture's source in 16-bit words (EncodePassPackLcu's packing, whole picture), pinned */
    SvtAmdMdLcu *md_lcus;
    SvtAmdOisLcuResult *md_ois;
    SvtAmdMeLcuResult *md_me;
    SvtAmdTmvpLcu *md_tmvp;
} EpPictureEntry;

typedef struct {
    union { /* the heads of the 8- and the 16-bit contract are the same */
        SvtAmdLcuWork work;
        SvtAmdLcuWork16 work16;
    };
    union {
        SvtAmdLcuResult res;
        SvtAmdLcuResult16 res16;
    };
    const LargestCodingUnit_t *lcu;
    int wide; /* 10-bit encode: the 16-bit members are live */
    int last_luma_cbf; /* the device's luma cbf of the transform unit the reference quantised last (EncodeTuCalcCost follows it) */
} EpServe;
_Static_assert(offsetof(SvtAmdLcuWork, src_y) == offsetof(SvtAmdLcuWork16, src_y) && offsetof(SvtAmdLcuResult, rec_y) == offsetof(SvtAmdLcuResult16, rec_y),
               "contract heads");
void EncodePassPackLcu(SequenceControlSet_t *sequenceControlSetPtr, EbPictureBufferDesc_t *inputPicture, EncDecContext_t *contextPtr,
                       EB_U32 lcuOriginX, EB_U32 lcuOriginY, EB_U32 lcuWidth, EB_U32 lcuHeight); /* EbCodingLoop.c:2867 */

__thread int svt_hook_ep_active;
static __thread EpServe *t_serve;
/* the LCU's records of the picture's device call (SVT_HOOK_MD) into the thread's staging */
static void serve_from_md(const EpPictureEntry *e, EB_U32 lcu)
{
    const size_t wb = e->wide ? sizeof(SvtAmdLcuWork16) : sizeof(SvtAmdLcuWork), rb = e->wide ? sizeof(SvtAmdLcuResult16) : sizeof(SvtAmdLcuResult);
    memcpy(&t_serve->work, (const uint8_t *)e->md_works + wb * lcu, wb);
    memcpy(&t_serve->res, (const uint8_t *)e->md_res + rb * lcu, rb);
    t_serve->wide = e->wide;
}
static __thread int t_md_kinds; /* the served LCU's work record comes from the device's mode decision: its inter_kind fields are decisions, not predictions */

static pthread_mutex_t g_ep_lock = PTHREAD_MUTEX_INITIALIZER; /* picture table + lane pool */
static pthread_cond_t g_ep_cv = PTHREAD_COND_INITIALIZER;
static EpPictureEntry g_ep_pic[EP_PICTURES];
static SvtAmdContext *g_ep_lane[EP_LANES];
static int g_ep_lane_busy[EP_LANES];
static int g_ep_state; /* 0 unknown, 1 on, -1 off */
static int g_ep_own;   /* SVT_HOOK_ENCODEPASS itself is set: LCUs of pictures the device's mode decision does not cover are encoded by per-LCU device calls;
                        * with SVT_HOOK_MD alone those pictures stay with the reference's own EncodePass */
static int g_ep_verify;  /* SVT_HOOK_ENCODEPASS_VERIFY: the reference encodes the LCU itself after the device call and the two outcomes are compared */
static int g_ep_refs;    /* SVT_HOOK_ENCODEPASS_REFS: pictures encoded entirely on the device are finished there and become reference pictures */
static unsigned long g_ep_refs_done, g_ep_refs_skipped;
static unsigned long g_ep_verified, g_ep_mismatch;
static unsigned long g_ep_gpu, g_ep_cpu_units, g_ep_cpu_tools, g_ep_cpu_format, g_ep_borders, g_ep_puts, g_ep_inter_units, g_ep_inter_lcus;

static __thread SvtAmdContext *t_lane; /* the lane this thread holds (svt_hook_die gives it back) */
static void lane_release(SvtAmdContext *lane);
void svt_hook_encdec_thread_exit(void)
{
    if (t_lane)
        lane_release(t_lane);
}
/* How many lanes there are: every lane is a stream, i.e. a hardware queue of the device, and the part schedules only so many at once - with the front half's lanes, the
 * root context and the runtime's own, 16 EncDec lanes made EVERY kernel of the process 20 - 25 % slower from the first launch on (profiles/r05_ai: mode-decision kernels
 * median 68 -> 90 ms at the same 10 - 12 launches side by side; r05_an: back to 75 ms with 12 + 3 lanes at pool 16).  SVT_HOOK_EP_LANES=<n> caps them (the closed-loop
 * configuration bench.py measures: 12, with SVT_HOOK_FRONT_LANES=4); a picture beyond the cap waits for a lane holding nothing on the device.  Without the switch: EP_LANES,
 * the count every end-to-end case of tests/ has run with. */
static int ep_lanes(void)
{
    static int n;
    if (!n) {
        const char *v = svt_hook_cfg("SVT_HOOK_EP_LANES");
        const int want = v ? atoi(v) : 0;
        n = (want >= 1 && want <= EP_LANES) ? want : EP_LANES;
    }
    return n;
}
static SvtAmdContext *lane_claim(SvtAmdContext *root)
{
    svt_hook_lock(&g_ep_lock);
    for (;;) {
        for (int i = 0; i < ep_lanes(); i++)
            if (!g_ep_lane_busy[i]) {
                if (!g_ep_lane[i] && svt_amd_context_fork(root, &g_ep_lane[i]))
                    svt_hook_die("svt_amd_context_fork (encode pass)");
                g_ep_lane_busy[i] = 1;
                svt_hook_unlock(&g_ep_lock);
                return t_lane = g_ep_lane[i];
            }
        pthread_cond_wait(&g_ep_cv, &g_ep_lock);
    }
}

static void lane_release(SvtAmdContext *lane)
{
    t_lane = NULL;
    svt_hook_lock(&g_ep_lock);
    for (int i = 0; i < EP_LANES; i++)
        if (g_ep_lane[i] == lane)
            g_ep_lane_busy[i] = 0;
    pthread_cond_signal(&g_ep_cv);
    svt_hook_unlock(&g_ep_lock);
}

/* everything an entry owns (under g_ep_lock) */
static void entry_release(SvtAmdContext *lane, EpPictureEntry *e)
{
    const int pins[2] = {e->ref_pins_plus1[0] - 1, e->ref_pins_plus1[1] - 1};
    svt_hook_release_references(pins);
    if (e->pic)
        svt_amd_encdec_picture_destroy(lane, e->pic);
    free(e->pending), free(e->works_all), free(e->res_all), free(e->sao_enable);
    void *pinned[] = {e->md_out, e->md_works, e->md_res, e->md_lcus, e->md_ois, e->md_me, e->md_tmvp, e->md_src16[0], e->md_src16[1], e->md_src16[2]};
    for (size_t i = 0; i < sizeof(pinned) / sizeof(pinned[0]); i++)
        if (pinned[i])
            svt_amd_host_free(lane, pinned[i]);
    pthread_mutex_destroy(&e->lock);
    memset(e, 0, sizeof(*e));
}

/* the last kernel thread of the encoder has returned (svt_hook_me.c:hook_teardown): picture objects and lanes go */
void svt_hook_encdec_teardown(void)
{
    svt_hook_lock(&g_ep_lock);
    SvtAmdContext *any = NULL;
    for (int i = 0; i < EP_LANES && !any; i++)
        any = g_ep_lane[i];
    for (int i = 0; i < EP_PICTURES; i++)
        if (g_ep_pic[i].pcs && any)
            entry_release(any, &g_ep_pic[i]);
    for (int i = 0; i < EP_LANES; i++) {
        if (g_ep_lane[i])
            svt_amd_context_destroy(g_ep_lane[i]);
        g_ep_lane[i] = NULL, g_ep_lane_busy[i] = 0;
    }
    svt_hook_unlock(&g_ep_lock);
}

/* the device picture of this PictureControlSet_t, begun for its current picture */
static void picture_prepare(SvtAmdContext *lane, EpPictureEntry *e, const PictureControlSet_t *pcs, int wide, int begin);
static double g_prep_t[3]; /* seconds inside picture_prepare: reference pictures resident, svt_amd_encdec_picture_set_inter, _begin (racy sums, a report only) */
static unsigned long g_prep_n;
/* The per-LCU wraps run 2 x 2,040 times per 4K picture on ~30 threads: the common case - the picture's ONE device call has returned, the LCU is answered from its records -
 * must not touch a global lock or hold a device lane (profiles/r04_q: with a lane claimed per call, eight pictures inside their device calls held every lane and all other
 * pictures' bookkeeping stood still; the global mutex alone cost ~36 ms per picture).  Entries are created once per PictureControlSet_t under g_ep_lock and never move. */
static EpPictureEntry *entry_lookup(const SequenceControlSet_t *scs, const PictureControlSet_t *pcs, int wide)
{
    for (int i = 0; i < EP_PICTURES; i++) {
        EpPictureEntry *e = &g_ep_pic[i];
        if (__atomic_load_n(&e->pcs, __ATOMIC_ACQUIRE) == pcs)
            return (e->wide == wide && e->width == scs->lumaWidth && e->height == scs->lumaHeight) ? e : NULL;
    }
    return NULL;
}
/* prepare == 0: only find (or create) the object; the caller decides whether the picture needs the device at all (SVT_HOOK_MD alone) and calls
 * picture_prepare itself under the entry's lock */
static EpPictureEntry *picture_entry_of(SvtAmdContext *lane, uint32_t lumaWidth, uint32_t lumaHeight, const PictureControlSet_t *pcs, int wide, int prepare);
static EpPictureEntry *picture_entry(SvtAmdContext *lane, const SequenceControlSet_t *scs, const PictureControlSet_t *pcs, int wide, int prepare)
{
    return picture_entry_of(lane, scs->lumaWidth, scs->lumaHeight, pcs, wide, prepare);
}
static EpPictureEntry *picture_entry_of(SvtAmdContext *lane, uint32_t lumaWidth, uint32_t lumaHeight, const PictureControlSet_t *pcs, int wide, int prepare)
{
    EpPictureEntry *e = NULL;
    svt_hook_lock(&g_ep_lock);
    for (int i = 0; i < EP_PICTURES && !e; i++)
        if (g_ep_pic[i].pcs == pcs)
            e = &g_ep_pic[i];
    const int ncap = (int)(((lumaWidth + 63u) / 64u) * ((lumaHeight + 63u) / 64u));
    if (e && (e->cap != ncap || e->wide != wide || e->width != lumaWidth || e->height != lumaHeight)) {
        /* a PictureControlSet_t address of an earlier encoder instance with another picture format: nothing of the entry fits */
        entry_release(lane, e);
        e = NULL;
    }
    for (int i = 0; i < EP_PICTURES && !e; i++)
        if (!g_ep_pic[i].pcs) {
            e = &g_ep_pic[i];
            pthread_mutex_init(&e->lock, NULL);
            e->wide = wide, e->width = lumaWidth, e->height = lumaHeight;
            if (svt_amd_encdec_picture_create(lane, (uint16_t)lumaWidth, (uint16_t)lumaHeight, wide ? 2 : 1, &e->pic))
                svt_hook_die("svt_amd_encdec_picture_create");
            e->cap = ncap;
            e->pending = malloc((wide ? sizeof(SvtAmdLcuBorder16) : sizeof(SvtAmdLcuBorder)) * (size_t)e->cap);
            if (!e->pending)
                svt_hook_die("out of memory (encode-pass border list)");
            __atomic_store_n(&e->pcs, pcs, __ATOMIC_RELEASE); /* published last: entry_lookup reads without the lock */
        }
    if (!e)
        svt_hook_die("encode pass: more PictureControlSet_t objects than EP_PICTURES");
    if (prepare)
        picture_prepare(lane, e, pcs, wide, 1);
    svt_hook_unlock(&g_ep_lock);
    return e;
}

/* first LCU of a new picture in this object: nothing coded yet (under g_ep_lock, or under the entry's lock when the caller holds that) */
static void picture_prepare(SvtAmdContext *lane, EpPictureEntry *e, const PictureControlSet_t *pcs, int wide, int begin)
{
    if (e->picture_plus1 == pcs->pictureNumber + 1)
        return;
    {   /* what the inter units of the picture read: reference pictures + rate tables (also what the PM-core quantiser of an I picture
         * prices its levels with); complete before other lanes launch: the begin below waits for this lane's stream */
        SvtAmdRefPicture refs[2];
        int have[2] = {0, 0};
        struct timespec t0, t1, t2, t3;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        {   /* the previous picture of this object no longer reads its reference pictures (its device work was waited for before its LCUs were served) */
            const int old[2] = {e->ref_pins_plus1[0] - 1, e->ref_pins_plus1[1] - 1};
            svt_hook_release_references(old);
            e->ref_pins_plus1[0] = e->ref_pins_plus1[1] = 0;
        }
        if (pcs->sliceType != EB_I_PICTURE) {
            int slot[2];
            svt_hook_resident_references(lane, pcs, wide, refs, have, slot); /* pinned in the cache for as long as this picture is in the object */
            e->ref_pins_plus1[0] = slot[0] + 1, e->ref_pins_plus1[1] = slot[1] + 1;
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (svt_amd_encdec_picture_set_inter(lane, e->pic, have[0] ? &refs[0] : NULL, have[1] ? &refs[1] : NULL, (const SvtAmdCabacCost *)pcs->cabacCost))
            svt_hook_die("svt_amd_encdec_picture_set_inter");
        clock_gettime(CLOCK_MONOTONIC, &t2);
        /* the picture-level mode-decision call resets the object itself (begin = 0): a second reset here costs a stream synchronisation behind whatever the
         * device is running - 12 ms per picture at BASELINE configs[2] with two pictures' kernels in flight */
        if (begin && svt_amd_encdec_picture_begin(lane, e->pic))
            svt_hook_die("svt_amd_encdec_picture_begin");
        clock_gettime(CLOCK_MONOTONIC, &t3);
        g_prep_t[0] += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        g_prep_t[1] += (double)(t2.tv_sec - t1.tv_sec) + 1e-9 * (double)(t2.tv_nsec - t1.tv_nsec);
        g_prep_t[2] += (double)(t3.tv_sec - t2.tv_sec) + 1e-9 * (double)(t3.tv_nsec - t2.tv_nsec);
        g_prep_n++;
    }
    e->picture_plus1 = pcs->pictureNumber + 1;
    e->npending = 0;
    e->sao_any = e->sao_varies = e->on_device = e->done = 0;
    if (g_ep_refs) {
        const size_t wb = wide ? sizeof(SvtAmdLcuWork16) : sizeof(SvtAmdLcuWork), rb = wide ? sizeof(SvtAmdLcuResult16) : sizeof(SvtAmdLcuResult);
        if (!e->works_all) {
            e->works_all = malloc(wb * (size_t)e->cap), e->res_all = malloc(rb * (size_t)e->cap), e->sao_enable = (uint8_t *)malloc((size_t)e->cap);
            if (!e->works_all || !e->res_all || !e->sao_enable)
                svt_hook_die("out of memory (encode-pass picture records)");
        }
        memset(e->sao_enable, 0, (size_t)e->cap);
    }
}

/* EncDec input contract: the coded leaves of the LCU in Z order.  Returns 0 when a unit is outside what the device call covers. */
static int fill_work(SvtAmdLcuWork *w, const SequenceControlSet_t *scs, const PictureControlSet_t *pcs, const LargestCodingUnit_t *lcuPtr,
                     EB_U32 lcuOriginX, EB_U32 lcuOriginY, const EncDecContext_t *contextPtr)
{
    memset(w, 0, offsetof(SvtAmdLcuWork, src_y));
    /* inter units: no coefficient shaping (encMode >= 11, EbEncDecProcess.c:2211) */
    const int inter_ok = !contextPtr->fastEl;
    int inter_units = 0;
    w->full_lambda = contextPtr->fullLambda; /* EncDecConfigureLcu ran before EncodePass (EbEncDecProcess.c:3004) */
    w->luma_cbf_bits[0] = contextPtr->mdRateEstimationPtr->lumaCbfBits[0], w->luma_cbf_bits[1] = contextPtr->mdRateEstimationPtr->lumaCbfBits[1];
    w->luma_cbf_bits[2] = contextPtr->mdRateEstimationPtr->lumaCbfBits[NUMBER_OF_CBF_CASES >> 1];
    w->luma_cbf_bits[3] = contextPtr->mdRateEstimationPtr->lumaCbfBits[(NUMBER_OF_CBF_CASES >> 1) + 1];
    w->lcu_x = (uint16_t)lcuOriginX, w->lcu_y = (uint16_t)lcuOriginY;
    w->slice_type = (uint8_t)pcs->sliceType, w->temporal_layer = pcs->temporalLayerIndex;
    w->constrained_intra = pcs->constrainedIntraFlag, w->strong_smoothing = scs->enableStrongIntraSmoothing;
    w->tile_left = lcuPtr->lcuEdgeInfoPtr->tileLeftEdgeFlag, w->tile_top = lcuPtr->lcuEdgeInfoPtr->tileTopEdgeFlag;
    w->tile_right = lcuPtr->lcuEdgeInfoPtr->tileRightEdgeFlag;
    w->pm_core = contextPtr->mdContext->rdoqPmCoreMethod == EB_PMCORE; /* encMode 1..4: the PM-core quantiser, on the device too */
    /* without the delta-QP tools every unit is coded at the picture's QP (EbCodingLoop.c:3214, :3238-3246) */
    const EB_U8 qp = pcs->pictureQp;
    const EB_U8 chromaQp = MapChromaQp((EB_U8)CLIP3((EB_S8)MIN_QP_VALUE, (EB_S8)MAX_CHROMA_MAP_QP_VALUE, (EB_S8)(qp + pcs->cbQpOffset + pcs->sliceCbQpOffset)));
    EB_U32 cuItr = 0, n = 0;
    while (cuItr < CU_MAX_COUNT) {
        const CodingUnit_t *cu = lcuPtr->codedLeafArrayPtr[cuItr];
        if (cu->splitFlag) {
            cuItr++;
            continue;
        }
        const CodedUnitStats_t *st = GetCodedUnitStats(cuItr);
        const int intra = cu->predictionModeFlag == INTRA_MODE;
        if (st->size < 8 || n >= SVT_AMD_LCU_MAX_CUS || (intra && (cu->predictionUnitArray->intraLumaMode == EB_INTRA_MODE_4x4 || st->size > 32)) ||
            (!intra && (!inter_ok || cu->predictionModeFlag != INTER_MODE)))
            return 0;
        SvtAmdLcuCu *u = &w->cu[n++];
        u->x = st->originX, u->y = st->originY, u->size = st->size, u->pred_mode = (uint8_t)cu->predictionModeFlag;
        u->intra_luma_mode = intra ? (uint8_t)cu->predictionUnitArray->intraLumaMode : 0;
        if (!intra) {
            const PredictionUnit_t *pu = cu->predictionUnitArray;
            u->inter_dir = (uint8_t)pu->interPredDirectionIndex;
            for (int l = 0; l < 2; l++)
                u->mv[l][0] = pu->mv[l].x, u->mv[l][1] = pu->mv[l].y;
            /* the merge / skip decision EncodePass is about to make (EbCodingLoop.c:3838-3882; isFirstCUinRow is false without the delta-QP
             * tools), on a copy of the cost it biases */
            u->inter_kind = SVT_AMD_EP_INTER_AMVP;
            if (pu->mergeFlag) {
                if (lcuPtr->chromaEncodeMode == CHROMA_MODE_BEST) {
                    /* the mode decision left chroma out of the merge / skip costs: EncodePass adds it first (:3840-3863, host code: chroma
                     * prediction + the chroma full loop of the merge candidate).  Done here with the same arguments, so that the decision
                     * is known before the device call; the call EncodePass makes later recomputes the same two costs. */
                    EbPictureBufferDesc_t *cin = pcs->ParentPcsPtr->chromaDownSamplePicturePtr;
                    const EB_U32 cuX = lcuOriginX + st->originX, cuY = lcuOriginY + st->originY;
                    ModeDecisionContext_t *md = contextPtr->mdContext;
                    md->cuOriginX = cuX, md->cuOriginY = cuY, md->puItr = 0, md->cuSize = st->size, md->cuSizeLog2 = st->sizeLog2, md->cuStats = st;
                    ((EncDecContext_t *)contextPtr)->cuStats = st;
                    AddChromaEncDec((PictureControlSet_t *)pcs, (LargestCodingUnit_t *)lcuPtr, (CodingUnit_t *)cu, md, (EncDecContext_t *)contextPtr, cin,
                                    ((cuY >> 1) + (cin->originY >> 1)) * cin->strideCb + ((cuX >> 1) + (cin->originX >> 1)),
                                    (((cuY & 63) * 32) + (cuX & 63)) >> 1, 0);
                }
                EB_U64 skipCost = contextPtr->mdContext->mdEpPipeLcu[cu->leafIndex].skipCost;
                if (pcs->sliceType == EB_B_PICTURE && pcs->ParentPcsPtr->isUsedAsReferenceFlag == EB_FALSE) {
                    static const EB_U8 INTRA_AREA_TH[MAX_TEMPORAL_LAYERS] = {40, 30, 30, 0, 0, 0};
                    const EbReferenceObject_t *r0 = (const EbReferenceObject_t *)pcs->refPicPtrArray[REF_LIST_0]->objectPtr;
                    const EbReferenceObject_t *r1 = (const EbReferenceObject_t *)pcs->refPicPtrArray[REF_LIST_1]->objectPtr;
                    if (pcs->ParentPcsPtr->variance[lcuPtr->index][0] < 200 &&
                        (r0->intraCodedArea > INTRA_AREA_TH[r0->tmpLayerIdx] || r1->intraCodedArea > INTRA_AREA_TH[r1->tmpLayerIdx]))
                        skipCost += (skipCost * 70) / 100;
                }
                u->inter_kind = skipCost <= contextPtr->mdContext->mdEpPipeLcu[cu->leafIndex].mergeCost ? SVT_AMD_EP_INTER_SKIP : SVT_AMD_EP_INTER_MERGE;
            }
            inter_units++;
        }
        const uint32_t lg = (uint32_t)Log2f(st->size);
        const uint32_t cuIndex = (st->originY >> lg) * (1u << st->depth) + (st->originX >> lg);
        u->bottom_left_ok = isBottomLeftAvailable(st->depth, cuIndex), u->top_right_ok = isUpperRightAvailable(st->depth, cuIndex);
        u->qp = qp, u->chroma_qp = chromaQp, u->leaf_index = (uint8_t)cuItr, u->dz_offset = 0;
        cuItr += DepthOffset[st->depth];
    }
    w->num_cus = (uint8_t)n;
    if (n > 0 && inter_units) {
        __atomic_add_fetch(&g_ep_inter_units, (unsigned long)inter_units, __ATOMIC_RELAXED);
        __atomic_add_fetch(&g_ep_inter_lcus, 1, __ATOMIC_RELAXED);
    }
    return n > 0;
}

/* what a host-encoded LCU leaves for its neighbours: the top / left entries of the ep* neighbour arrays over its extent */
static void border_from_neighbour_arrays(void *out, int wide, const PictureControlSet_t *pcs, EB_U32 tileIdx, EB_U32 x0, EB_U32 y0, EB_U32 lw,
                                         EB_U32 lh)
{
    NeighborArrayUnit_t *mode = pcs->epModeTypeNeighborArray[tileIdx];
    NeighborArrayUnit_t *na[3] = {wide ? pcs->epLumaReconNeighborArray16bit[tileIdx] : pcs->epLumaReconNeighborArray[tileIdx],
                                  wide ? pcs->epCbReconNeighborArray16bit[tileIdx] : pcs->epCbReconNeighborArray[tileIdx],
                                  wide ? pcs->epCrReconNeighborArray16bit[tileIdx] : pcs->epCrReconNeighborArray[tileIdx]};
    uint8_t *mb, *mr;
    void *dst[6]; /* bottom / right of Y, Cb, Cr */
    if (wide) {
        SvtAmdLcuBorder16 *b = (SvtAmdLcuBorder16 *)out;
        memset(b, 0, sizeof(*b));
        b->lcu_x = (uint16_t)x0, b->lcu_y = (uint16_t)y0, mb = b->mode_bottom, mr = b->mode_right;
        dst[0] = b->bottom_y, dst[1] = b->right_y, dst[2] = b->bottom_cb, dst[3] = b->right_cb, dst[4] = b->bottom_cr, dst[5] = b->right_cr;
    } else {
        SvtAmdLcuBorder *b = (SvtAmdLcuBorder *)out;
        memset(b, 0, sizeof(*b));
        b->lcu_x = (uint16_t)x0, b->lcu_y = (uint16_t)y0, mb = b->mode_bottom, mr = b->mode_right;
        dst[0] = b->bottom_y, dst[1] = b->right_y, dst[2] = b->bottom_cb, dst[3] = b->right_cb, dst[4] = b->bottom_cr, dst[5] = b->right_cr;
    }
    for (EB_U32 k = 0; k < lw / 4; k++)
        mb[k] = mode->topArray[GetNeighborArrayUnitTopIndex(mode, x0 + 4 * k)];
    for (EB_U32 k = 0; k < lh / 4; k++)
        mr[k] = mode->leftArray[GetNeighborArrayUnitLeftIndex(mode, y0 + 4 * k)];
    const size_t bps = wide ? 2 : 1;
    for (int p = 0; p < 3; p++) {
        const EB_U32 sh = p ? 1 : 0;
        memcpy(dst[2 * p], na[p]->topArray + (size_t)(x0 >> sh) * bps, (size_t)(lw >> sh) * bps);
        memcpy(dst[2 * p + 1], na[p]->leftArray + (size_t)(y0 >> sh) * bps, (size_t)(lh >> sh) * bps);
    }
}

/* the SAO wraps (svt_hook_me.c) report every decision call made while an LCU is being served */
void svt_hook_ep_note_sao(const PictureControlSet_t *pcs, EB_U32 x, EB_U32 y, const MdRateEstimationContext_t *md, EB_U64 lambda, EB_U64 chromaLambda,
                          int mmSao, int is16)
{
    if (!g_ep_refs)
        return;
    EpPictureEntry *e = NULL;
    svt_hook_lock(&g_ep_lock);
    for (int i = 0; i < EP_PICTURES && !e; i++)
        if (g_ep_pic[i].pcs == pcs)
            e = &g_ep_pic[i];
    svt_hook_unlock(&g_ep_lock);
    if (!e || !e->sao_enable)
        return;
    SvtAmdSaoDecisionParams P;
    memset(&P, 0, sizeof(P));
    P.lambda = lambda, P.chroma_lambda = chromaLambda;
    for (int k = 0; k < 6; k++)
        P.type_bits[k] = md->saoTypeIndexBits[k];
    for (int k = 0; k < 2; k++)
        P.merge_bits[k] = md->saoMergeFlagBits[k];
    for (int k = 0; k < 8; k++)
        P.offset_bits[k] = md->saoOffsetTrunUnaryBits[k];
    P.is_10bit = (uint8_t)is16, P.mm_sao = mmSao ? 1 : 0, P.temporal_layer = pcs->temporalLayerIndex;
    const SequenceControlSet_t *scs = (const SequenceControlSet_t *)pcs->ParentPcsPtr->sequenceControlSetWrapperPtr->objectPtr;
    const EB_U32 wl = (scs->lumaWidth + 63u) / 64u;
    svt_hook_lock(&e->lock);
    if (!e->sao_any)
        e->sao_P = P, e->sao_any = 1;
    else if (memcmp(&e->sao_P, &P, sizeof(P)))
        e->sao_varies = 1; /* per-LCU lambdas (QP modulation): the picture-level decision call does not cover it */
    e->sao_enable[(y / 64u) * wl + x / 64u] = 1;
    svt_hook_unlock(&e->lock);
}

/* the last LCU of a picture is through and every LCU went through the device: finish the picture there (what EncDecKernel does on the host
 * afterwards, EbEncDecProcess.c:3040-3200: remaining deblocking, ApplySaoOffsetsPicture, PadRefAndSetFlags) and hand it to the reference cache */
static void finish_picture_on_device(SvtAmdContext *lane, EpPictureEntry *e, const SequenceControlSet_t *scs, const PictureControlSet_t *pcs, int mismatch)
{
    const int wide = e->wide;
    if (!pcs->ParentPcsPtr->isUsedAsReferenceFlag)
        return;
    if (e->on_device != e->cap || e->sao_varies || (scs->staticConfig.disableDlfFlag && scs->staticConfig.enableSaoFlag)) {
        __atomic_add_fetch(&g_ep_refs_skipped, 1, __ATOMIC_RELAXED);
        return;
    }
    const EbReferenceObject_t *ro = (const EbReferenceObject_t *)pcs->ParentPcsPtr->referencePictureWrapperPtr->objectPtr;
    const EbPictureBufferDesc_t *rp = wide ? ro->referencePicture16bit : ro->referencePicture;
    /* allowEncDecMismatch (EbEncDecProcess.c:2036-2054: temporal layers > 0 at encMode >= 8, and at encMode 7 in 4K): the encoder neither
     * deblocks (EbCodingLoop.c:3081) nor applies SAO (EbEncDecProcess.c:3085) on its side - its reference picture is the padded
     * reconstruction as encoded */
    if (!scs->staticConfig.disableDlfFlag && !mismatch) {
        SvtAmdDeblockParams prm;
        memset(&prm, 0, sizeof(prm));
        prm.tc_offset = pcs->tcOffset, prm.beta_offset = pcs->betaOffset, prm.cb_qp_offset = pcs->cbQpOffset, prm.cr_qp_offset = pcs->crQpOffset;
        prm.slice_type = (uint8_t)pcs->sliceType;
        prm.ref_poc[0] = prm.ref_poc[1] = ~0ull;
        for (int l = 0; l < (pcs->sliceType == EB_B_PICTURE ? 2 : pcs->sliceType == EB_P_PICTURE ? 1 : 0); l++)
            prm.ref_poc[l] = ((const EbReferenceObject_t *)pcs->refPicPtrArray[l]->objectPtr)->refPOC;
        if (wide ? svt_amd_encdec_picture_deblock16(lane, e->pic, (const SvtAmdLcuWork16 *)e->works_all, (const SvtAmdLcuResult16 *)e->res_all, &prm, NULL, NULL, NULL)
                 : svt_amd_encdec_picture_deblock(lane, e->pic, (const SvtAmdLcuWork *)e->works_all, (const SvtAmdLcuResult *)e->res_all, &prm, NULL, NULL, NULL))
            svt_hook_die("svt_amd_encdec_picture_deblock");
        if (scs->staticConfig.enableSaoFlag && e->sao_any &&
            (wide ? svt_amd_encdec_picture_sao16(lane, e->pic, (const SvtAmdLcuWork16 *)e->works_all, &e->sao_P, e->sao_enable, NULL, NULL, NULL, NULL)
                  : svt_amd_encdec_picture_sao(lane, e->pic, (const SvtAmdLcuWork *)e->works_all, &e->sao_P, e->sao_enable, NULL, NULL, NULL, NULL)))
            svt_hook_die("svt_amd_encdec_picture_sao");
    }
    SvtAmdRefPicture dev;
    if (svt_amd_encdec_picture_reference(lane, e->pic, rp->originX, rp->originY, &dev, NULL, NULL, NULL))
        svt_hook_die("svt_amd_encdec_picture_reference");
    svt_hook_register_device_reference(rp, pcs->pictureNumber, wide ? 2 : 1, &dev);
    __atomic_add_fetch(&g_ep_refs_done, 1, __ATOMIC_RELAXED);
}

/* one more LCU of the picture is through EncodePass (served from the device or not); the thread that brings the last one finishes the picture */
static void picture_lcu_done(SvtAmdContext *root, EpPictureEntry *e, const SequenceControlSet_t *scs, const PictureControlSet_t *pcs, EB_U32 tbAddr, int served,
                             int mismatch)
{
    if (!g_ep_refs)
        return;
    svt_hook_lock(&e->lock);
    if (served) {
        const size_t wb = e->wide ? sizeof(SvtAmdLcuWork16) : sizeof(SvtAmdLcuWork), rb = e->wide ? sizeof(SvtAmdLcuResult16) : sizeof(SvtAmdLcuResult);
        memcpy((uint8_t *)e->works_all + wb * tbAddr, &t_serve->work, wb);
        memcpy((uint8_t *)e->res_all + rb * tbAddr, &t_serve->res, rb);
        e->on_device++;
    }
    const int last = ++e->done == e->cap;
    svt_hook_unlock(&e->lock);
    if (last) {
        SvtAmdContext *lane = lane_claim(root);
        finish_picture_on_device(lane, e, scs, pcs, mismatch);
        lane_release(lane);
    }
}

/* Verification mode: the outcome of the reference's own EncodePass of the LCU (flags, coefficients and - with the loop filters off - the
 * reconstruction) against what the device returned for it */
static void verify_lcu(const SequenceControlSet_t *scs, const PictureControlSet_t *pcs, const LargestCodingUnit_t *lcuPtr, EB_U32 x0, EB_U32 y0,
                       int limit_intra)
{
    const SvtAmdLcuWork *w = &t_serve->work;
    const int wide = t_serve->wide;
    const EbPictureBufferDesc_t *q = lcuPtr->quantizedCoeff;
    const int16_t *dq[3] = {t_serve->res.coeff_y, t_serve->res.coeff_cb, t_serve->res.coeff_cr};
    const EB_S16 *rq[3] = {(const EB_S16 *)q->bufferY, (const EB_S16 *)q->bufferCb, (const EB_S16 *)q->bufferCr};
    const EbPictureBufferDesc_t *rec = pcs->ParentPcsPtr->isUsedAsReferenceFlag
                                           ? (wide ? ((EbReferenceObject_t *)pcs->ParentPcsPtr->referencePictureWrapperPtr->objectPtr)->referencePicture16bit
                                                   : ((EbReferenceObject_t *)pcs->ParentPcsPtr->referencePictureWrapperPtr->objectPtr)->referencePicture)
                                           : (wide ? pcs->reconPicture16bitPtr : pcs->reconPicturePtr);
    /* the reconstruction is comparable with the loop filters off, where the reference produced one (doRecon, EbCodingLoop.c:3083-3087) */
    int any_intra = 0;
    for (int i = 0; i < w->num_cus; i++)
        any_intra |= w->cu[i].pred_mode == INTRA_MODE;
    const int rec_ok = scs->staticConfig.disableDlfFlag && !scs->staticConfig.enableSaoFlag &&
                       (!limit_intra || any_intra || pcs->ParentPcsPtr->isUsedAsReferenceFlag || scs->staticConfig.reconEnabled);
    const size_t bps = wide ? 2 : 1;
    const uint8_t *dr[3] = {(const uint8_t *)t_serve->res.rec_y, wide ? (const uint8_t *)t_serve->res16.rec_cb : t_serve->res.rec_cb,
                            wide ? (const uint8_t *)t_serve->res16.rec_cr : t_serve->res.rec_cr};
    const EB_U8 *rr[3] = {rec->bufferY, rec->bufferCb, rec->bufferCr};
    const EB_U32 rs[3] = {rec->strideY, rec->strideCb, rec->strideCr};
    int bad = 0;
    for (int i = 0; i < w->num_cus; i++) {
        const SvtAmdLcuCu *u = &w->cu[i];
        const CodingUnit_t *cu = lcuPtr->codedLeafArrayPtr[u->leaf_index];
        const int skip = u->pred_mode == INTER_MODE && u->inter_kind == SVT_AMD_EP_INTER_SKIP;
        const int ntu = u->size == 64 ? 4 : 1, T = u->size == 64 ? 32 : u->size;
        for (int tu = 0; tu < ntu; tu++) {
            const TransformUnit_t *t = &cu->transformUnitArray[u->size == 64 ? 1 + tu : 0];
            const SvtAmdLcuCuResult *d = &t_serve->res.cu[i + (u->size == 64 ? 1 + tu : 0)];
            const int tx = u->x + ((tu & 1) << 5), ty = u->y + ((tu >> 1) << 5);
            const int rcbf[3] = {t->lumaCbf, t->cbCbf, t->crCbf};
            for (int p = 0; p < 3; p++) {
                const int sh = p ? 1 : 0, n = T >> sh, pitch = p ? 32 : 64, lx = tx >> sh, ly = ty >> sh;
                int what = 0;
                if (d->cbf[p] != rcbf[p])
                    what |= 1;
                if (!skip && (d->nz[p] != t->nzCoefCount[p] || d->only_dc[p] != t->isOnlyDc[p]))
                    what |= 2;
                for (int r = 0; r < n && !skip && !(what & 4); r++)
                    if (memcmp(dq[p] + (ly + r) * pitch + lx, rq[p] + (ly + r) * pitch + lx, (size_t)n * 2))
                        what |= 4;
                for (int r = 0; r < n && rec_ok && !(what & 8); r++)
                    if (memcmp(dr[p] + ((size_t)(ly + r) * pitch + lx) * bps,
                               rr[p] + ((size_t)(((rec->originY + y0) >> sh) + ly + r) * rs[p] + ((rec->originX + x0) >> sh) + lx) * bps, (size_t)n * bps))
                        what |= 8;
                if (what && bad++ < 4)
                    fprintf(stderr,
                            "svt_hook_encdec: VERIFY picture %llu slice %d lcu (%u,%u) unit %d (%d,%d) size %d mode %d kind %d dir %d mv (%d,%d)(%d,%d) tu %d plane %d: "
                            "%s%s%s%s cbf dev %d ref %d nz dev %d ref %d dc dev %d ref %d skipFlag %d mergeFlag %d\n",
                            (unsigned long long)pcs->pictureNumber, (int)pcs->sliceType, x0, y0, i, u->x, u->y, u->size, u->pred_mode, u->inter_kind, u->inter_dir,
                            u->mv[0][0], u->mv[0][1], u->mv[1][0], u->mv[1][1], tu, p, what & 1 ? "CBF " : "", what & 2 ? "COUNT " : "", what & 4 ? "COEFF " : "",
                            what & 8 ? "REC " : "", d->cbf[p], rcbf[p], d->nz[p], t->nzCoefCount[p], d->only_dc[p], t->isOnlyDc[p], (int)cu->skipFlag,
                            (int)cu->predictionUnitArray->mergeFlag);
            }
        }
    }
    __atomic_add_fetch(&g_ep_verified, 1, __ATOMIC_RELAXED);
    if (bad)
        __atomic_add_fetch(&g_ep_mismatch, 1, __ATOMIC_RELAXED);
}

/* the served pass re-derives the merge / skip decision of every inter unit (EbCodingLoop.c:3838-3882, :4139, :4347-4352) from the
 * same costs fill_work read.  Should the two ever disagree the bitstream would signal one thing and the reconstruction hold another: check
 * the flags EncodePass left against what the device encoded, and stop loudly instead of drifting. */
static void check_inter_kinds(const LargestCodingUnit_t *lcuPtr)
{
    const SvtAmdLcuWork *w = &t_serve->work;
    for (int i = 0; i < w->num_cus; i++) {
        const SvtAmdLcuCu *u = &w->cu[i];
        if (u->pred_mode != INTER_MODE)
            continue;
        const CodingUnit_t *cu = lcuPtr->codedLeafArrayPtr[u->leaf_index];
        const int merge = cu->predictionUnitArray->mergeFlag, skip = cu->skipFlag;
        int coded = 0;
        for (int k = 0; k < (u->size == 64 ? 5 : 1); k++)
            coded |= t_serve->res.cu[i + k].cbf[0] | t_serve->res.cu[i + k].cbf[1] | t_serve->res.cu[i + k].cbf[2];
        /* a unit decided skip leaves EncodePass with skipFlag set and mergeFlag CLEARED (:4139-4140); a merge unit keeps mergeFlag and is
         * forced to skip only when nothing was coded (:4348-4352); an AMVP unit has neither */
        const int ok = u->inter_kind == SVT_AMD_EP_INTER_SKIP ? (skip && !merge)
                     : u->inter_kind == SVT_AMD_EP_INTER_MERGE ? (merge && (!skip || !coded)) : (!merge && !skip);
        if (!ok)
            svt_hook_die("encode pass: the reference's merge / skip decision of an inter unit differs from what the device encoded");
    }
}

/* AddChromaEncDec (Codec/EbProductCodingLoop.c:4158): EncodePass completes the merge / skip costs of a merge unit of a CHROMA_MODE_BEST LCU with
 * chroma before it decides (EbCodingLoop.c:3840-3880).  When the LCU was decided AND encoded by the device (svt_amd_md_encode_picture_inter) that
 * decision is part of the work record: the two costs are set so that the pass reaches it (the 70 % skip-cost bias that may follow keeps the order). */
void __real_AddChromaEncDec(PictureControlSet_t *pictureControlSetPtr, LargestCodingUnit_t *lcuPtr, CodingUnit_t *cuPtr, ModeDecisionContext_t *contextPtr,
                            EncDecContext_t *contextPtrED, EbPictureBufferDesc_t *inputPicturePtr, EB_U32 inputCbOriginIndex, EB_U32 cuChromaOriginIndex,
                            EB_U32 candIdxInput);
void __wrap_AddChromaEncDec(PictureControlSet_t *pictureControlSetPtr, LargestCodingUnit_t *lcuPtr, CodingUnit_t *cuPtr, ModeDecisionContext_t *contextPtr,
                            EncDecContext_t *contextPtrED, EbPictureBufferDesc_t *inputPicturePtr, EB_U32 inputCbOriginIndex, EB_U32 cuChromaOriginIndex,
                            EB_U32 candIdxInput)
{
    if (!(svt_hook_ep_active && t_md_kinds)) {
        __real_AddChromaEncDec(pictureControlSetPtr, lcuPtr, cuPtr, contextPtr, contextPtrED, inputPicturePtr, inputCbOriginIndex, cuChromaOriginIndex, candIdxInput);
        return;
    }
    const SvtAmdLcuWork *w = &t_serve->work;
    for (int i = 0; i < w->num_cus; i++)
        if (w->cu[i].leaf_index == cuPtr->leafIndex) {
            const int skip = w->cu[i].inter_kind == SVT_AMD_EP_INTER_SKIP;
            contextPtr->mdEpPipeLcu[cuPtr->leafIndex].skipCost = skip ? 0 : 1, contextPtr->mdEpPipeLcu[cuPtr->leafIndex].mergeCost = skip ? 1 : 0;
            return;
        }
    svt_hook_die("mode decision: EncodePass asks for the merge / skip costs of a unit the device did not decide");
}

static void watchdog_tick(void); /* SVT_HOOK_WATCHDOG, below */
void __wrap_EncodePass(SequenceControlSet_t *scs, PictureControlSet_t *pcs, LargestCodingUnit_t *lcuPtr, EB_U32 tbAddr, EB_U32 lcuOriginX,
                       EB_U32 lcuOriginY, EB_U32 lcuQp, EB_BOOL enableSaoFlag, EncDecContext_t *contextPtr)
{
    if (g_ep_state == 0) {
        g_ep_verify = getenv("SVT_HOOK_ENCODEPASS_VERIFY") != NULL;
        g_ep_refs = getenv("SVT_HOOK_ENCODEPASS_REFS") != NULL;
        g_ep_state = (getenv("SVT_HOOK_ENCODEPASS") || svt_hook_cfg("SVT_HOOK_MD")) ? 1 : -1;
        g_ep_own = getenv("SVT_HOOK_ENCODEPASS") != NULL;
    }
    if (g_ep_state < 0 || svt_hook_failed() || contextPtr->colorFormat != EB_YUV420 || (scs->lumaWidth & 7) || (scs->lumaHeight & 7)) {
        if (g_ep_state > 0)
            __atomic_add_fetch(&g_ep_cpu_format, 1, __ATOMIC_RELAXED);
        __real_EncodePass(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
        return;
    }
    svt_hook_note_callback(scs);
    watchdog_tick();
    const int wide = contextPtr->is16bit != 0; /* 10-bit encode: 16-bit samples, EncodeLoop16bit */
    {   /* the common case of SVT_HOOK_MD: the picture's device call has returned - answered from its records without a global lock or a device lane (see entry_lookup) */
        EpPictureEntry *f = entry_lookup(scs, pcs, wide);
        if (f && __atomic_load_n(&f->md_done_plus1, __ATOMIC_ACQUIRE) == pcs->pictureNumber + 1 && f->md_picture_plus1 == pcs->pictureNumber + 1) {
            const int tools0 = scs->staticConfig.improveSharpness || scs->staticConfig.bitRateReduction || scs->staticConfig.segmentOvEnabled ||
                               (contextPtr->mdContext->rdoqPmCoreMethod != EB_NO_RDOQ && contextPtr->mdContext->rdoqPmCoreMethod != EB_PMCORE);
            if (f->md_ok && !tools0) {
                if (!t_serve && !(t_serve = (EpServe *)malloc(sizeof(EpServe))))
                    svt_hook_die("out of memory (encode-pass staging)");
                if (svt_hook_timeline_enabled()) {
                    svt_hook_lock(&f->lock);
                    if (f->tl_lcus == 0)
                        f->tl_first = svt_hook_now();
                    const int all = ++f->tl_lcus == f->cap;
                    const double first = f->tl_first;
                    if (all)
                        f->tl_lcus = 0;
                    svt_hook_unlock(&f->lock);
                    if (all)
                        svt_hook_timeline("encodepass", pcs->pictureNumber, pcs->temporalLayerIndex, (int)pcs->sliceType, first, svt_hook_now());
                }
                serve_from_md(f, tbAddr);
                __atomic_add_fetch(&g_ep_gpu, 1, __ATOMIC_RELAXED);
                t_serve->lcu = lcuPtr;
                svt_hook_ep_active = 1, t_md_kinds = 1;
                __real_EncodePass(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
                svt_hook_ep_active = 0, t_md_kinds = 0;
                check_inter_kinds(lcuPtr);
                if (g_ep_refs)
                    picture_lcu_done(svt_hook_device((uint16_t)scs->lumaWidth, (uint16_t)scs->lumaHeight), f, scs, pcs, tbAddr, 1, contextPtr->allowEncDecMismatch);
                return;
            }
            if (!f->md_ok && !g_ep_own) { /* SVT_HOOK_MD alone, a picture outside the device's mode decision: the reference code's, EncodePass included */
                __atomic_add_fetch(&g_ep_cpu_units, 1, __ATOMIC_RELAXED);
                __real_EncodePass(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
                if (svt_hook_timeline_enabled()) { /* the host's own pictures on the same time line: first counted LCU .. last LCU through EncodePass (the first LCUs of a
                                                     * picture whose leading LCUs take the BDP path pass before the picture's entry exists and are not counted) */
                    svt_hook_lock(&f->lock);
                    if (f->tl_lcus == 0)
                        f->tl_first = svt_hook_now();
                    f->tl_lcus++;
                    const double first = f->tl_first;
                    const int last = tbAddr + 1 == (EB_U32)f->cap;
                    if (last)
                        f->tl_lcus = 0;
                    svt_hook_unlock(&f->lock);
                    if (last)
                        svt_hook_timeline("encodepass", pcs->pictureNumber, pcs->temporalLayerIndex, (int)pcs->sliceType, first, svt_hook_now());
                }
                return;
            }
        }
    }
    SvtAmdContext *root = svt_hook_device((uint16_t)scs->lumaWidth, (uint16_t)scs->lumaHeight);
    SvtAmdContext *lane = lane_claim(root);
    EpPictureEntry *e = picture_entry(lane, scs, pcs, wide, g_ep_own);
    const EB_U32 lw = MIN(64u, scs->lumaWidth - lcuOriginX), lh = MIN(64u, scs->lumaHeight - lcuOriginY);
    if (svt_hook_timeline_enabled()) { /* SVT_HOOK_TIMELINE: first / last LCU of the picture through EncodePass */
        svt_hook_lock(&e->lock);
        if (e->tl_lcus == 0)
            e->tl_first = svt_hook_now();
        const int all = ++e->tl_lcus == e->cap;
        const double first = e->tl_first;
        if (all)
            e->tl_lcus = 0;
        svt_hook_unlock(&e->lock);
        if (all)
            svt_hook_timeline("encodepass", pcs->pictureNumber, pcs->temporalLayerIndex, (int)pcs->sliceType, first, svt_hook_now());
    }
    if (!t_serve && !(t_serve = (EpServe *)malloc(sizeof(EpServe))))
        svt_hook_die("out of memory (encode-pass staging)");
    /* tools that change a unit's QP, dead zone, coefficient shape or quantiser are outside this revision */
    const int tools = scs->staticConfig.improveSharpness || scs->staticConfig.bitRateReduction || scs->staticConfig.segmentOvEnabled ||
                      (contextPtr->mdContext->rdoqPmCoreMethod != EB_NO_RDOQ && contextPtr->mdContext->rdoqPmCoreMethod != EB_PMCORE); /* RDOQ: encMode 0 */
    const int md_served = !tools && e->md_ok && e->md_picture_plus1 == pcs->pictureNumber + 1;
    if (md_served) { /* the picture's device call (svt_hook_md_lcu) encoded this LCU already: no work record to build, no device call to make */
        serve_from_md(e, tbAddr);
        lane_release(lane);
        __atomic_add_fetch(&g_ep_gpu, 1, __ATOMIC_RELAXED);
        t_serve->lcu = lcuPtr;
        svt_hook_ep_active = 1, t_md_kinds = 1; /* the device made the merge / skip decisions too: AddChromaEncDec is answered from the work record */
        __real_EncodePass(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
        svt_hook_ep_active = 0, t_md_kinds = 0;
        check_inter_kinds(lcuPtr);
        picture_lcu_done(root, e, scs, pcs, tbAddr, 1, contextPtr->allowEncDecMismatch);
        return;
    }
    if (!g_ep_own) { /* SVT_HOOK_MD alone: a picture outside the device's mode decision is the reference code's, EncodePass included */
        lane_release(lane);
        __atomic_add_fetch(&g_ep_cpu_units, 1, __ATOMIC_RELAXED);
        __real_EncodePass(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
        return;
    }
    const int units = !tools && fill_work(&t_serve->work, scs, pcs, lcuPtr, lcuOriginX, lcuOriginY, contextPtr);
    if (!units) {
        lane_release(lane);
        __atomic_add_fetch(tools ? &g_ep_cpu_tools : &g_ep_cpu_units, 1, __ATOMIC_RELAXED);
        __real_EncodePass(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
        svt_hook_lock(&e->lock);
        if (e->npending >= e->cap)
            svt_hook_die("encode pass: border list overflow");
        border_from_neighbour_arrays((uint8_t *)e->pending + (size_t)e->npending++ * (wide ? sizeof(SvtAmdLcuBorder16) : sizeof(SvtAmdLcuBorder)), wide,
                                     pcs, contextPtr->encDecTileIndex, lcuOriginX, lcuOriginY, lw, lh);
        __atomic_add_fetch(&g_ep_borders, 1, __ATOMIC_RELAXED);
        svt_hook_unlock(&e->lock);
        picture_lcu_done(root, e, scs, pcs, tbAddr, 0, contextPtr->allowEncDecMismatch);
        return;
    }
    /* source samples of the LCU */
    EbPictureBufferDesc_t *in = (EbPictureBufferDesc_t *)pcs->ParentPcsPtr->enhancedPicturePtr;
    SvtAmdLcuWork *w = &t_serve->work;
    t_serve->wide = wide;
    if (wide) { /* the reference packs the 8 + 2 bit planes of the LCU into its 16-bit LCU buffer at the top of EncodePass; do it now */
        EncodePassPackLcu(scs, in, contextPtr, lcuOriginX, lcuOriginY, lw, lh);
        const EbPictureBufferDesc_t *s16 = contextPtr->inputSample16bitBuffer;
        SvtAmdLcuWork16 *w16 = &t_serve->work16;
        for (EB_U32 y = 0; y < lh; y++)
            memcpy(w16->src_y + y * 64, (const uint16_t *)s16->bufferY + (size_t)y * s16->strideY, lw * 2);
        for (EB_U32 y = 0; y < lh / 2; y++) {
            memcpy(w16->src_cb + y * 32, (const uint16_t *)s16->bufferCb + (size_t)y * s16->strideCb, lw);
            memcpy(w16->src_cr + y * 32, (const uint16_t *)s16->bufferCr + (size_t)y * s16->strideCr, lw);
        }
    } else {
        for (EB_U32 y = 0; y < lh; y++)
            memcpy(w->src_y + y * 64, in->bufferY + (size_t)(in->originY + lcuOriginY + y) * in->strideY + in->originX + lcuOriginX, lw);
        for (EB_U32 y = 0; y < lh / 2; y++) {
            memcpy(w->src_cb + y * 32, in->bufferCb + (size_t)((in->originY + lcuOriginY) / 2 + y) * in->strideCb + (in->originX + lcuOriginX) / 2, lw / 2);
            memcpy(w->src_cr + y * 32, in->bufferCr + (size_t)((in->originY + lcuOriginY) / 2 + y) * in->strideCr + (in->originX + lcuOriginX) / 2, lw / 2);
        }
    }
    /* LCUs the host encoded since the last device call enter the device picture first (under the picture's lock, so that a second
     * thread's LCU cannot overtake a put it depends on) */
    svt_hook_lock(&e->lock);
    if (e->npending) {
        if (wide ? svt_amd_encdec_picture_put_borders16(lane, e->pic, (const SvtAmdLcuBorder16 *)e->pending, e->npending)
                 : svt_amd_encdec_picture_put_borders(lane, e->pic, (const SvtAmdLcuBorder *)e->pending, e->npending))
            svt_hook_die("svt_amd_encdec_picture_put_borders");
        e->npending = 0;
        __atomic_add_fetch(&g_ep_puts, 1, __ATOMIC_RELAXED);
    }
    svt_hook_unlock(&e->lock);
    if (wide ? svt_amd_encode_lcus16(lane, e->pic, &t_serve->work16, 1, &t_serve->res16) : svt_amd_encode_lcus(lane, e->pic, w, 1, &t_serve->res))
        svt_hook_die("svt_amd_encode_lcus");
    lane_release(lane);
    __atomic_add_fetch(&g_ep_gpu, 1, __ATOMIC_RELAXED);
    t_serve->lcu = lcuPtr;
    if (g_ep_verify) { /* the reference encodes the LCU itself; the device's outcome is only compared */
        __real_EncodePass(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
        verify_lcu(scs, pcs, lcuPtr, lcuOriginX, lcuOriginY, contextPtr->mdContext->limitIntra);
        return;
    }
    svt_hook_ep_active = 1;
    __real_EncodePass(scs, pcs, lcuPtr, tbAddr, lcuOriginX, lcuOriginY, lcuQp, enableSaoFlag, contextPtr);
    svt_hook_ep_active = 0;
    check_inter_kinds(lcuPtr);
    picture_lcu_done(root, e, scs, pcs, tbAddr, 1, contextPtr->allowEncDecMismatch);
}

void __wrap_PictureResidual(EB_U8 *input, EB_U32 inputStride, EB_U8 *pred, EB_U32 predStride, EB_S16 *residual, EB_U32 residualStride,
                            EB_U32 areaWidth, EB_U32 areaHeight)
{
    if (!svt_hook_ep_active)
        __real_PictureResidual(input, inputStride, pred, predStride, residual, residualStride, areaWidth, areaHeight);
}

void __real_PictureResidual16bit(EB_U16 *input, EB_U32 inputStride, EB_U16 *pred, EB_U32 predStride, EB_S16 *residual, EB_U32 residualStride,
                                 EB_U32 areaWidth, EB_U32 areaHeight);
void __wrap_PictureResidual16bit(EB_U16 *input, EB_U32 inputStride, EB_U16 *pred, EB_U32 predStride, EB_S16 *residual, EB_U32 residualStride,
                                 EB_U32 areaWidth, EB_U32 areaHeight)
{
    if (!svt_hook_ep_active)
        __real_PictureResidual16bit(input, inputStride, pred, predStride, residual, residualStride, areaWidth, areaHeight);
}

/* EncodeLoop16bit's transform (EbCodingLoop.c:1321; the C table of EncodeTransform holds the same Estimate forms for 10-bit) */
EB_ERRORTYPE __real_EncodeTransform(EB_S16 *residualBuffer, EB_U32 residualStride, EB_S16 *coeffBuffer, EB_U32 coeffStride, EB_U32 transformSize,
                                    EB_S16 *transformInnerArrayPtr, EB_U32 bitIncrement, EB_BOOL dstTransformFlag, EB_TRANS_COEFF_SHAPE transCoeffShape);
EB_ERRORTYPE __wrap_EncodeTransform(EB_S16 *residualBuffer, EB_U32 residualStride, EB_S16 *coeffBuffer, EB_U32 coeffStride, EB_U32 transformSize,
                                    EB_S16 *transformInnerArrayPtr, EB_U32 bitIncrement, EB_BOOL dstTransformFlag, EB_TRANS_COEFF_SHAPE transCoeffShape)
{
    if (svt_hook_ep_active)
        return EB_ErrorNone;
    return __real_EncodeTransform(residualBuffer, residualStride, coeffBuffer, coeffStride, transformSize, transformInnerArrayPtr, bitIncrement,
                                  dstTransformFlag, transCoeffShape);
}

EB_ERRORTYPE __wrap_EstimateTransform(EB_S16 *residualBuffer, EB_U32 residualStride, EB_S16 *coeffBuffer, EB_U32 coeffStride, EB_U32 transformSize,
                                      EB_S16 *transformInnerArrayPtr, EB_U32 bitIncrement, EB_BOOL dstTansformFlag,
                                      EB_TRANS_COEFF_SHAPE transCoeffShape)
{
    if (svt_hook_ep_active)
        return EB_ErrorNone;
    return __real_EstimateTransform(residualBuffer, residualStride, coeffBuffer, coeffStride, transformSize, transformInnerArrayPtr, bitIncrement,
                                    dstTansformFlag, transCoeffShape);
}

/* the unit of the served LCU the reference is at */
static int serve_unit(const EncDecContext_t *contextPtr)
{
    const SvtAmdLcuWork *w = &t_serve->work;
    const EB_U32 x = contextPtr->cuStats->originX, y = contextPtr->cuStats->originY;
    for (int i = 0; i < w->num_cus; i++)
        if (w->cu[i].x == x && w->cu[i].y == y && w->cu[i].size == contextPtr->cuStats->size)
            return i;
    svt_hook_die("encode pass: the reference reached a unit the device call did not hold");
    return -1;
}

void svt_hook_ep_quantize(EncDecContext_t *contextPtr, EB_S16 *quantCoeff, EB_S16 *reconCoeff, EB_U32 coeffStride, EB_U32 qp, EB_U32 areaSize,
                          EB_U32 *nz, EB_U32 shape, EB_U32 cleanSparse, EB_U32 masking, EB_U32 enableCbflag, EB_U32 contouring, EB_U32 dZoffset)
{
    const int i = serve_unit(contextPtr);
    const SvtAmdLcuCu *u = &t_serve->work.cu[i];
    const EbPictureBufferDesc_t *q = t_serve->lcu->quantizedCoeff;
    /* which plane and which transform unit of the unit: the destination lies in one of the LCU's three coefficient planes (64x64 units
     * have four 32x32 transform units, result entries i + 1 .. i + 4) */
    int p = -1;
    const EB_S16 *base[3] = {(const EB_S16 *)q->bufferY, (const EB_S16 *)q->bufferCb, (const EB_S16 *)q->bufferCr};
    EB_U32 tx = 0, ty = 0;
    for (int k = 0; k < 3; k++) {
        const EB_U32 pitch = k ? 32 : 64;
        const ptrdiff_t d = quantCoeff - base[k];
        if (d >= 0 && d < (ptrdiff_t)(pitch * pitch))
            p = k, tx = (EB_U32)d % pitch, ty = (EB_U32)d / pitch;
    }
    const EB_U32 sh = p > 0 ? 1 : 0, T = u->size == 64 ? 32u : u->size, n = T >> sh;
    if (p >= 0) /* luma coordinates inside the LCU */
        tx <<= sh, ty <<= sh;
    const int inside = p >= 0 && tx >= u->x && ty >= u->y && tx < (EB_U32)u->x + u->size && ty < (EB_U32)u->y + u->size && !((tx - u->x) & (T - 1)) &&
                       !((ty - u->y) & (T - 1));
    if (!inside || areaSize != n || coeffStride != (p ? 32u : 64u) || qp != (EB_U32)(p ? u->chroma_qp : u->qp) + (t_serve->wide ? 12u : 0u) || shape ||
        cleanSparse || masking || enableCbflag || contouring || dZoffset || !nz ||
        (u->pred_mode == INTER_MODE && u->inter_kind == SVT_AMD_EP_INTER_SKIP))
        svt_hook_die("encode pass: the reference's quantiser call differs from what the device encoded (unit, plane, QP, skip decision or tool flags)");
    const int e = i + (u->size == 64 ? 1 + (int)(((ty - u->y) >> 5) * 2 + ((tx - u->x) >> 5)) : 0);
    const int16_t *src = (p == 0 ? t_serve->res.coeff_y : p == 1 ? t_serve->res.coeff_cb : t_serve->res.coeff_cr) + (quantCoeff - base[p]);
    for (EB_U32 r = 0; r < n; r++)
        memcpy(quantCoeff + r * coeffStride, src + r * coeffStride, n * sizeof(int16_t));
    *nz = t_serve->res.cu[e].nz[p];
    t_serve->last_luma_cbf = p == 0 ? t_serve->res.cu[e].cbf[0] : t_serve->last_luma_cbf;
    reconCoeff[0] = t_serve->res.cu[e].only_dc[p]; /* the caller's isOnlyDc test (EbCodingLoop.c:792, 879, 1000) reads the DC term */
}

void svt_hook_ep_recon(EncDecContext_t *contextPtr, EB_U32 originX, EB_U32 originY, EB_U32 tuSize, EbPictureBufferDesc_t *recon)
{
    const int i = serve_unit(contextPtr);
    const SvtAmdLcuCu *u = &t_serve->work.cu[i];
    const EB_U32 ux = originX & 63, uy = originY & 63; /* the transform unit inside the LCU (64x64 units: four of 32) */
    if (tuSize != (u->size == 64 ? 32u : u->size) || ux < u->x || uy < u->y || ux + tuSize > (EB_U32)u->x + u->size || uy + tuSize > (EB_U32)u->y + u->size)
        svt_hook_die("encode pass: reconstruction call for another unit");
    const size_t bps = t_serve->wide ? 2 : 1;
    const uint8_t *ry = (const uint8_t *)t_serve->res.rec_y, *rcb, *rcr; /* rec_y sits at the same offset in both contracts */
    if (t_serve->wide)
        rcb = (const uint8_t *)t_serve->res16.rec_cb, rcr = (const uint8_t *)t_serve->res16.rec_cr;
    else
        rcb = t_serve->res.rec_cb, rcr = t_serve->res.rec_cr;
    EB_U8 *y = recon->bufferY + ((size_t)(recon->originY + originY) * recon->strideY + recon->originX + originX) * bps;
    for (EB_U32 r = 0; r < tuSize; r++)
        memcpy(y + (size_t)r * recon->strideY * bps, ry + ((size_t)(uy + r) * 64 + ux) * bps, tuSize * bps);
    EB_U8 *cb = recon->bufferCb + ((size_t)((recon->originY + originY) / 2) * recon->strideCb + (recon->originX + originX) / 2) * bps;
    EB_U8 *cr = recon->bufferCr + ((size_t)((recon->originY + originY) / 2) * recon->strideCr + (recon->originX + originX) / 2) * bps;
    for (EB_U32 r = 0; r < tuSize / 2; r++) {
        memcpy(cb + (size_t)r * recon->strideCb * bps, rcb + ((size_t)(uy / 2 + r) * 32 + ux / 2) * bps, tuSize / 2 * bps);
        memcpy(cr + (size_t)r * recon->strideCr * bps, rcr + ((size_t)(uy / 2 + r) * 32 + ux / 2) * bps, tuSize / 2 * bps);
    }
}

/* The luma cbf decision of an AMVP unit (EbCodingLoop.c:4075-4124) was made on the device: the two measuring leaves return at once, and
 * EncodeTuCalcCost - which also writes the TransformUnit_t flags - is given distortions that reproduce the device's decision. */
EB_ERRORTYPE __wrap_PictureFullDistortionLuma(EbPictureBufferDesc_t *coeff, EB_U32 coeffLumaOriginIndex, EbPictureBufferDesc_t *reconCoeff,
                                              EB_U32 reconCoeffLumaOriginIndex, EB_U32 areaSize, EB_U64 lumaDistortion[DIST_CALC_TOTAL],
                                              EB_U32 countNonZeroCoeffsY, EB_MODETYPE mode)
{
    if (svt_hook_ep_active) {
        lumaDistortion[0] = lumaDistortion[1] = 0;
        return EB_ErrorNone;
    }
    return __real_PictureFullDistortionLuma(coeff, coeffLumaOriginIndex, reconCoeff, reconCoeffLumaOriginIndex, areaSize, lumaDistortion, countNonZeroCoeffsY,
                                            mode);
}

EB_ERRORTYPE __wrap_TuEstimateCoeffBitsEncDec(EB_U32 tuOriginIndex, EB_U32 tuChromaOriginIndex, EntropyCoder_t *entropyCoderPtr,
                                              EbPictureBufferDesc_t *coeffBufferTB, EB_U32 countNonZeroCoeffs[3], EB_U64 *yTuCoeffBits,
                                              EB_U64 *cbTuCoeffBits, EB_U64 *crTuCoeffBits, EB_U32 transformSize, EB_U32 transformChromaSize,
                                              EB_MODETYPE type, CabacCost_t *CabacCost)
{
    if (svt_hook_ep_active)
        return EB_ErrorNone; /* the caller zeroed the three counters (:4097-4099) */
    return __real_TuEstimateCoeffBitsEncDec(tuOriginIndex, tuChromaOriginIndex, entropyCoderPtr, coeffBufferTB, countNonZeroCoeffs, yTuCoeffBits,
                                            cbTuCoeffBits, crTuCoeffBits, transformSize, transformChromaSize, type, CabacCost);
}

EB_ERRORTYPE __wrap_EncodeTuCalcCost(EncDecContext_t *contextPtr, EB_U32 *countNonZeroCoeffs, EB_U64 yTuDistortion[DIST_CALC_TOTAL], EB_U64 *yTuCoeffBits,
                                     EB_U32 componentMask)
{
    if (svt_hook_ep_active) { /* keep: coded cost 0 < zeroed cost; drop: the other way round (the rates only add to both) */
        const int keep = t_serve->last_luma_cbf;
        yTuDistortion[DIST_CALC_RESIDUAL] = keep ? 0 : (EB_U64)1 << 40;
        yTuDistortion[DIST_CALC_PREDICTION] = keep ? (EB_U64)1 << 40 : 0;
        *yTuCoeffBits = 0;
    }
    return __real_EncodeTuCalcCost(contextPtr, countNonZeroCoeffs, yTuDistortion, yTuCoeffBits, componentMask);
}


/* ---- SVT_HOOK_MD: ModeDecisionLcu (Codec/EbProductCodingLoop.c:4691) answered from ONE device call per picture -------------------- */
/* SVT_HOOK_WATCHDOG=<seconds>: a thread that ends the process LOUDLY when no LCU has passed EncodePass for that long while pictures are inside the bindings - with
 * what every picture object, lane and the device's launch budget hold at that moment.  A diagnosis tool for the bench and the sweeps (a wedged encoder must say why
 * instead of eating the run's time limit); off unless asked for. */
static unsigned long g_wd_progress;
static int g_wd_state; /* 0 unknown, 1 running, -1 off */
static void *watchdog_main(void *arg)
{
    const int limit = (int)(intptr_t)arg;
    unsigned long last = __atomic_load_n(&g_wd_progress, __ATOMIC_RELAXED);
    int idle = 0;
    for (;;) {
        struct timespec ts = {1, 0};
        nanosleep(&ts, NULL);
        const unsigned long now = __atomic_load_n(&g_wd_progress, __ATOMIC_RELAXED);
        int busy = 0;
        for (int i = 0; i < EP_PICTURES; i++) {
            const EpPictureEntry *e = &g_ep_pic[i];
            if (e->pcs && e->md_picture_plus1 && e->md_done_plus1 != e->md_picture_plus1)
                busy++;
        }
        int fl = 0, wg = 0, wt = 0;
        (void)svt_amd_debug_md_flights(&fl, &wg, &wt);
        idle = (now == last && (busy || fl || wt)) ? idle + 1 : 0;
        last = now;
        if (idle < limit)
            continue;
        fprintf(stderr, "svt_hook_encdec: WATCHDOG: no LCU through EncodePass for %d s; device launches in flight %d holding %d workgroups, %d calls waiting\n", limit, fl, wg, wt);
        for (int i = 0; i < EP_LANES; i++)
            if (g_ep_lane_busy[i])
                fprintf(stderr, "    lane %d busy\n", i);
        for (int i = 0; i < EP_PICTURES; i++) {
            const EpPictureEntry *e = &g_ep_pic[i];
            if (e->pcs)
                fprintf(stderr, "    picture object %d: picture %llu (tl %d, slice %d), md call for %llu, returned for %llu, ok %d, LCUs served %d\n", i,
                        (unsigned long long)e->pcs->pictureNumber, (int)e->pcs->temporalLayerIndex, (int)e->pcs->sliceType,
                        (unsigned long long)e->md_picture_plus1 - 1, (unsigned long long)e->md_done_plus1 - 1, e->md_ok, e->tl_lcus);
        }
        fflush(stderr);
        abort();
    }
    return NULL;
}
static void watchdog_tick(void)
{
    __atomic_add_fetch(&g_wd_progress, 1, __ATOMIC_RELAXED);
    if (__atomic_load_n(&g_wd_state, __ATOMIC_ACQUIRE) == 0) {
        int expect = 0;
        const char *v = getenv("SVT_HOOK_WATCHDOG");
        const int secs = v ? atoi(v) : 0;
        if (__atomic_compare_exchange_n(&g_wd_state, &expect, secs > 0 ? 1 : -1, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE) && secs > 0) {
            pthread_t th;
            if (pthread_create(&th, NULL, watchdog_main, (void *)(intptr_t)secs) == 0)
                pthread_detach(th);
        }
    }
}

/* The EncDec picture pool.  The reference sizes it for host latencies: pictureControlSetPoolInitCountChild = MAX(4, coreCount / 6) PictureControlSet_t objects
 * between the picture manager and packetization (Codec/EbEncHandle.c:1801, built at :818-829) - five at -lp 32.  A picture the device decides and encodes stays there
 * for its device call (50 - 120 ms at 4K where the host's wavefront takes 25 - 40), so the closed loop's rate is (objects in the pool) / (residence) and the pool, not the
 * device, sets it.  SVT_HOOK_PCS_POOL=<n> raises the count of THAT pool (recognised by its creator) to n without touching the thread counts -lp also sets;
 * the bitstream does not depend on it (pictures are still released in decode order with their references complete).  The one-line change a maintainer makes in
 * LoadDefaultBufferConfigurationSettings is shown in INTEGRATION.md 1h. */
#include "EbSystemResourceManager.h"
/* First uses cost: a lane is a stream to create (20 - 30 ms while kernels run), a picture object and its mode-decision state are device allocations, the records of its
 * device calls page-locked host memory - and any of them made while other pictures' kernels run waits for those (profiles/r05_aa).  The EncDec pictures are known when
 * their pool is built (the binding below notes them); once the device is up - still inside EbInitEncoder, before the encode clock - every one of them gets its lane,
 * picture object, state and records.  SVT_HOOK_NO_WARMUP=1 leaves all of it to the first pictures. */
static void md_entry_buffers(SvtAmdContext *lane, EpPictureEntry *e);
static const PictureControlSet_t *g_warm_pcs[EP_PICTURES];
static int g_warm_n, g_warm_wide;
static uint32_t g_warm_w, g_warm_h;
void svt_hook_encdec_warmup(void)
{
    if (!g_warm_n || !svt_hook_cfg("SVT_HOOK_MD") || getenv("SVT_HOOK_NO_WARMUP") || (g_warm_w & 7) || (g_warm_h & 7))
        return;
    SvtAmdContext *root = svt_hook_device((uint16_t)g_warm_w, (uint16_t)g_warm_h);
    const int n = g_warm_n < ep_lanes() ? g_warm_n : ep_lanes();
    SvtAmdContext *lanes[EP_LANES];
    for (int i = 0; i < n; i++) /* all at once: a lane given back would be handed out again */
        lanes[i] = lane_claim(root);
    for (int i = 0; i < g_warm_n; i++) {
        SvtAmdContext *lane = lanes[i % n];
        EpPictureEntry *e = picture_entry_of(lane, g_warm_w, g_warm_h, g_warm_pcs[i], g_warm_wide, 0);
        md_entry_buffers(lane, e);
        if (svt_amd_md_picture_warmup(lane, e->pic))
            svt_hook_die("svt_amd_md_picture_warmup");
    }
    for (int i = 0; i < n; i++)
        lane_release(lanes[i]);
    g_warm_n = 0;
}
EB_ERRORTYPE EbInputBufferHeaderCreator(EB_PTR *objectDblPtr, EB_PTR objectInitDataPtr); /* Codec/EbEncHandle.c:3902 (no header declares it) */
EB_ERRORTYPE __real_EbSystemResourceCtor(EbSystemResource_t *resourcePtr, EB_U32 objectTotalCount, EB_U32 producerProcessTotalCount, EB_U32 consumerProcessTotalCount,
                                         EbFifo_t ***producerFifoPtrArrayPtr, EbFifo_t ***consumerFifoPtrArrayPtr, EB_BOOL fullFifoEnabled, EB_CREATOR objectCreator,
                                         EB_PTR objectInitDataPtr, EbDctor objectDestroyer);
EB_ERRORTYPE __wrap_EbSystemResourceCtor(EbSystemResource_t *resourcePtr, EB_U32 objectTotalCount, EB_U32 producerProcessTotalCount, EB_U32 consumerProcessTotalCount,
                                         EbFifo_t ***producerFifoPtrArrayPtr, EbFifo_t ***consumerFifoPtrArrayPtr, EB_BOOL fullFifoEnabled, EB_CREATOR objectCreator,
                                         EB_PTR objectInitDataPtr, EbDctor objectDestroyer)
{
    if (objectCreator == PictureControlSetCreator) {
        const char *v = svt_hook_cfg("SVT_HOOK_PCS_POOL");
        const int want = v ? atoi(v) : 0;
        if (want > (int)objectTotalCount && want <= EP_PICTURES) {
            fprintf(stderr, "svt_hook_encdec: EncDec picture pool %u -> %d PictureControlSet_t objects (SVT_HOOK_PCS_POOL)\n", objectTotalCount, want);
            objectTotalCount = (EB_U32)want;
        }
    }
    const EB_ERRORTYPE rc = __real_EbSystemResourceCtor(resourcePtr, objectTotalCount, producerProcessTotalCount, consumerProcessTotalCount, producerFifoPtrArrayPtr,
                                                        consumerFifoPtrArrayPtr, fullFifoEnabled, objectCreator, objectInitDataPtr, objectDestroyer);
    if (rc == EB_ErrorNone && objectCreator == PictureControlSetCreator && objectInitDataPtr) { /* svt_hook_encdec_warmup below */
        const PictureControlSetInitData_t *init = (const PictureControlSetInitData_t *)objectInitDataPtr;
        g_warm_n = 0, g_warm_w = init->pictureWidth, g_warm_h = init->pictureHeight, g_warm_wide = init->is16bit ? 1 : 0;
        for (EB_U32 i = 0; i < resourcePtr->objectTotalCount && i < EP_PICTURES; i++)
            g_warm_pcs[g_warm_n++] = (const PictureControlSet_t *)resourcePtr->wrapperPtrPool[i]->objectPtr;
    }
    /* the pools whose pictures cross PCIe when the mode decision runs on the device - source pictures (the application's input buffers the encoder keeps as
     * enhancedPicturePtr) and reference pictures - are page-locked HERE, while the encoder is being built and the device is idle (svt_hook_me.c:svt_hook_pin_picture_at_init) */
    if (rc == EB_ErrorNone && (svt_hook_cfg("SVT_HOOK_MD") || getenv("SVT_HOOK_ENCODEPASS")) && (objectCreator == EbInputBufferHeaderCreator || objectCreator == EbReferenceObjectCreator))
        for (EB_U32 i = 0; i < resourcePtr->objectTotalCount; i++) {
            void *obj = resourcePtr->wrapperPtrPool[i]->objectPtr;
            if (objectCreator == EbInputBufferHeaderCreator) {
                svt_hook_pin_picture_at_init((const EbPictureBufferDesc_t *)((EB_BUFFERHEADERTYPE *)obj)->pBuffer, 1);
            } else {
                const EbReferenceObject_t *r = (const EbReferenceObject_t *)obj;
                svt_hook_pin_picture_at_init(r->referencePicture, 1);
                svt_hook_pin_picture_at_init(r->referencePicture16bit, 2);
            }
        }
    return rc;
}

EB_ERRORTYPE __real_ModeDecisionLcu(SequenceControlSet_t *scs, PictureControlSet_t *pcs, const MdcLcuData_t *const mdcResultTbPtr,
                                    LargestCodingUnit_t *lcuPtr, EB_U16 lcuOriginX, EB_U16 lcuOriginY, EB_U32 lcuAddr, ModeDecisionContext_t *contextPtr);
static int g_md_state; /* 0 unknown, 1 on, -1 off */
static int g_md_skip_intra;
static int g_md_verify; /* SVT_HOOK_MD_VERIFY: the reference decides the LCU itself as well and the two trees are compared */
static unsigned long g_md_pictures, g_md_inter_pictures, g_md_lcus, g_md_left_pictures, g_md_verified, g_md_mismatch;

/* what ProductFullModeDecision (Codec/EbModeDecision.c:1995-2183) and the loop around it leave in the LCU's coding-unit array, from the
 * device's decision record */
static void md_apply(LargestCodingUnit_t *lcuPtr, const SvtAmdMdLcuOut *o, EB_U8 qp, ModeDecisionContext_t *md)
{
    for (int i = 0; i < SVT_AMD_MD_LEAVES; i++) {
        CodingUnit_t *cu = lcuPtr->codedLeafArrayPtr[i];
        cu->splitFlag = o->split[i];
        if (!o->tested[i])
            continue;
        /* mdEpPipeLcu[].mergeCost / .skipCost (EbModeDecision.c:2043-2044): in a CHROMA_MODE_FULL LCU EncodePass takes its merge / skip decision straight from these
         * (EbCodingLoop.c:3840-3873; in a CHROMA_MODE_BEST LCU AddChromaEncDec - wrapped above - completes them first) */
        md->mdEpPipeLcu[i].mergeCost = o->merge_cost[i], md->mdEpPipeLcu[i].skipCost = o->skip_cost[i];
        cu->leafIndex = (EB_U8)i, cu->qp = qp;
        cu->predictionModeFlag = o->pred_mode[i], cu->skipFlag = EB_FALSE, cu->rootCbf = o->ycbf[i] ? EB_TRUE : EB_FALSE;
        PredictionUnit_t *pu = cu->predictionUnitArray;
        pu->intraLumaMode = o->pred_mode[i] == INTRA_MODE ? o->intra_luma_mode[i] : 0x1F;
        pu->interPredDirectionIndex = o->inter_dir[i], pu->mergeFlag = o->merge_flag[i] ? EB_TRUE : EB_FALSE, pu->mergeIndex = o->merge_index[i];
        for (int l = 0; l < 2; l++)
            pu->mv[l].x = o->mv[i][l][0], pu->mv[l].y = o->mv[i][l][1];
        pu->mvd[REF_LIST_0].predIdx = pu->mvd[REF_LIST_1].predIdx = 0;
        TransformUnit_t *tu = &cu->transformUnitArray[0];
        if (i == 0) { /* a 64x64 unit: four 32x32 transform units; "exclude chroma from cost calculation" (EbProductCodingLoop.c:4460) leaves the
                       * candidate's chroma cbf at bit 0 only */
            tu->splitFlag = EB_TRUE, tu->cbCbf = tu->crCbf = EB_FALSE, tu->cbCbf2 = tu->crCbf2 = EB_FALSE, tu->chromaCbfContext = 0;
            for (int k = 1; k <= 4; k++) {
                TransformUnit_t *t4 = &cu->transformUnitArray[k];
                t4->splitFlag = EB_FALSE, t4->lumaCbf = (o->ycbf[i] >> k) & 1 ? EB_TRUE : EB_FALSE;
                t4->cbCbf = t4->crCbf = t4->cbCbf2 = t4->crCbf2 = EB_FALSE, t4->chromaCbfContext = 1, t4->lumaCbfContext = 0;
            }
            continue;
        }
        tu->splitFlag = EB_FALSE, tu->lumaCbf = o->ycbf[i] ? EB_TRUE : EB_FALSE;
        tu->cbCbf = tu->crCbf = EB_TRUE; /* candidatePtr->cbCbf = crCbf = 1 */
        tu->cbCbf2 = tu->crCbf2 = EB_FALSE, tu->chromaCbfContext = 0, tu->lumaCbfContext = 1;
    }
}

/* the page-locked records of an entry's device calls (first use, or the warm-up at EbInitEncoder time) */
static void md_entry_buffers(SvtAmdContext *lane, EpPictureEntry *e)
{
    if (e->md_out)
        return;
    const size_t n = (size_t)e->cap;
    const size_t wb = e->wide ? sizeof(SvtAmdLcuWork16) : sizeof(SvtAmdLcuWork), rb = e->wide ? sizeof(SvtAmdLcuResult16) : sizeof(SvtAmdLcuResult);
    if (svt_amd_host_alloc(lane, sizeof(SvtAmdMdLcuOut) * n, (void **)&e->md_out) || svt_amd_host_alloc(lane, wb * n, &e->md_works) ||
        svt_amd_host_alloc(lane, rb * n, &e->md_res) || svt_amd_host_alloc(lane, sizeof(SvtAmdMdLcu) * n, (void **)&e->md_lcus) ||
        svt_amd_host_alloc(lane, sizeof(SvtAmdOisLcuResult) * n, (void **)&e->md_ois))
        svt_hook_die("out of memory (mode-decision picture records)");
    if (e->wide)
        for (int k = 0; k < 3; k++)
            if (svt_amd_host_alloc(lane, sizeof(uint16_t) * (size_t)(e->width >> (k ? 1 : 0)) * (e->height >> (k ? 1 : 0)), (void **)&e->md_src16[k]))
                svt_hook_die("out of memory (10-bit source picture)");
}
/* the picture's ONE device call; under e->lock */
static double g_md_t_fill, g_md_t_call, g_md_t_prepare; /* seconds of host work / of device calls / of picture_prepare inside md_picture, summed (under the entries' locks: racy sums, a report only) */
static double md_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
/* picture-level part of the question "does this picture take the device call": needs no lane (a host-decided base-layer picture must not wait for one - the lanes are
 * held by the pictures inside their 60 - 100 ms device calls, and the base layer is the encoder's serial chain) */
static int md_picture_level_ok(SequenceControlSet_t *scs, PictureControlSet_t *pcs, ModeDecisionContext_t *md, SvtAmdMdPicture *P, SvtAmdMdInter *X)
{
    const int tools = scs->staticConfig.improveSharpness || scs->staticConfig.bitRateReduction || scs->staticConfig.segmentOvEnabled ||
                      scs->staticConfig.rateControlMode != 0; /* per-LCU QP / lambda: not in SvtAmdMdPicture */
    svt_md_fill_picture(P, scs, pcs, md);
    const int inter = pcs->sliceType != EB_I_PICTURE;
    if (inter)
        svt_md_fill_inter(X, scs, pcs, md);
    return !(tools || (!inter && g_md_skip_intra) || !(inter ? svt_amd_md_picture_supported_inter(P, X) : svt_amd_md_picture_supported(P)));
}
static void md_picture(SvtAmdContext *lane, EpPictureEntry *e, SequenceControlSet_t *scs, PictureControlSet_t *pcs, ModeDecisionContext_t *md)
{
    const double t_in = md_now();
    double t_dev = 0;
    e->md_picture_plus1 = pcs->pictureNumber + 1, e->md_ok = 0;
    SvtAmdMdPicture P;
    SvtAmdMdInter X;
    const int inter = pcs->sliceType != EB_I_PICTURE;
    if (!md_picture_level_ok(scs, pcs, md, &P, &X)) {
        __atomic_add_fetch(&g_md_left_pictures, 1, __ATOMIC_RELAXED);
        svt_hook_timeline("md_host", pcs->pictureNumber, pcs->temporalLayerIndex, (int)pcs->sliceType, t_in, md_now());
        return;
    }
    const size_t n = (size_t)e->cap;
    md_entry_buffers(lane, e);
    SvtAmdMdLcu *lcus = e->md_lcus;
    SvtAmdOisLcuResult *ois = e->md_ois;
    for (size_t i = 0; i < n; i++) {
        svt_md_fill_lcu(&lcus[i], scs, pcs, pcs->lcuPtrArray[i], md);
        svt_md_fill_ois(&ois[i], pcs->ParentPcsPtr, (EB_U32)i);
    }
    const EbPictureBufferDesc_t *in = pcs->ParentPcsPtr->chromaDownSamplePicturePtr;
    svt_hook_pin_picture(in, 1); /* the source planes go up by DMA from where the encoder keeps them */
    const uint8_t *sy = in->bufferY + (size_t)in->originY * in->strideY + in->originX;
    const uint8_t *scb = in->bufferCb + (size_t)(in->originY / 2) * in->strideCb + in->originX / 2, *scr = in->bufferCr + (size_t)(in->originY / 2) * in->strideCr + in->originX / 2;
    if (e->wide && (!inter || svt_amd_md_lcus_supported(&P, lcus, (int)n))) {
        /* a 10-bit picture: the source EncodePass codes is the 8-bit planes plus the two extra bits, packed LCU by LCU at the top of EncodePass (EncodePassPackLcu,
         * EbCodingLoop.c:2867) - here the reference's own packers (CompressedPackLcu / Pack2D_SRC, Codec/EbPictureOperators.h:160, :170) fill the whole picture at once;
         * the device derives the mode decision's 8-bit view from it */
        EbPictureBufferDesc_t *ip = (EbPictureBufferDesc_t *)pcs->ParentPcsPtr->enhancedPicturePtr;
        const EB_U32 W = scs->lumaWidth, H = scs->lumaHeight;
        for (EB_U32 y0 = 0; y0 < H; y0 += 64)
            for (EB_U32 x0 = 0; x0 < W; x0 += 64) {
                const EB_U32 lw = MIN(64u, W - x0), lh = MIN(64u, H - y0);
                uint16_t *dy = e->md_src16[0] + (size_t)y0 * W + x0, *dcb = e->md_src16[1] + (size_t)(y0 / 2) * (W / 2) + x0 / 2, *dcr = e->md_src16[2] + (size_t)(y0 / 2) * (W / 2) + x0 / 2;
                const EB_U32 oy = (y0 + ip->originY) * ip->strideY + x0 + ip->originX, ocb = ((y0 + ip->originY) >> 1) * ip->strideCb + ((x0 + ip->originX) >> 1),
                             ocr = ((y0 + ip->originY) >> 1) * ip->strideCr + ((x0 + ip->originX) >> 1);
                if (scs->staticConfig.compressedTenBitFormat == 1) {
                    const EB_U16 l2 = ip->width / 4, c2 = (ip->width / 4) >> 1;
                    CompressedPackLcu(ip->bufferY + oy, ip->strideY, ip->bufferBitIncY + y0 * l2 + (x0 / 4) * lh, lw / 4, dy, W, lw, lh);
                    CompressedPackLcu(ip->bufferCb + ocb, ip->strideCb, ip->bufferBitIncCb + (y0 >> 1) * c2 + ((x0 >> 1) / 4) * (lh >> 1), (lw >> 1) / 4, dcb, W / 2, lw >> 1, lh >> 1);
                    CompressedPackLcu(ip->bufferCr + ocr, ip->strideCr, ip->bufferBitIncCr + (y0 >> 1) * c2 + ((x0 >> 1) / 4) * (lh >> 1), (lw >> 1) / 4, dcr, W / 2, lw >> 1, lh >> 1);
                } else {
                    Pack2D_SRC(ip->bufferY + oy, ip->strideY, ip->bufferBitIncY + (y0 + ip->originY) * ip->strideBitIncY + x0 + ip->originX, ip->strideBitIncY, dy, W, lw, lh);
                    Pack2D_SRC(ip->bufferCb + ocb, ip->strideCr, ip->bufferBitIncCb + ((y0 + ip->originY) >> 1) * ip->strideBitIncCb + ((x0 + ip->originX) >> 1),
                               ip->strideBitIncCr, dcb, W / 2, lw >> 1, lh >> 1);
                    Pack2D_SRC(ip->bufferCr + ocr, ip->strideCr, ip->bufferBitIncCr + ((y0 + ip->originY) >> 1) * ip->strideBitIncCr + ((x0 + ip->originX) >> 1),
                               ip->strideBitIncCr, dcr, W / 2, lw >> 1, lh >> 1);
                }
            }
    }
    const double t_p0 = md_now();
    if (!inter || svt_amd_md_lcus_supported(&P, lcus, (int)n))
        picture_prepare(lane, e, pcs, e->wide, 0); /* the picture is the device's: reference pictures resident, rate tables (the call below resets the object) */
    const double t_c0 = md_now();
    g_md_t_prepare += t_c0 - t_p0;
    const EB_U32 W2 = scs->lumaWidth;
    if (!inter) {
        if (e->wide ? svt_amd_md_encode_picture16(lane, e->pic, &P, lcus, e->md_src16[0], W2, e->md_src16[1], e->md_src16[2], W2 / 2, ois, 0,
                                                  (const SvtAmdCabacCost *)pcs->cabacCost, e->md_out, (SvtAmdLcuWork16 *)e->md_works, (SvtAmdLcuResult16 *)e->md_res)
                    : svt_amd_md_encode_picture(lane, e->pic, &P, lcus, sy, in->strideY, scb, scr, in->strideCb, ois, 0, (const SvtAmdCabacCost *)pcs->cabacCost, e->md_out,
                                                (SvtAmdLcuWork *)e->md_works, (SvtAmdLcuResult *)e->md_res))
            svt_hook_die("svt_amd_md_encode_picture");
    } else {
        /* a P / B picture: every LCU must be one ModeDecisionLcu decides with luma-only candidates; the motion-estimation results and the
         * co-located picture's motion field go up with the call, the reference pictures are resident (picture_entry) */
        if (!svt_amd_md_lcus_supported(&P, lcus, (int)n)) {
            __atomic_add_fetch(&g_md_left_pictures, 1, __ATOMIC_RELAXED);
            svt_hook_timeline("md_host", pcs->pictureNumber, pcs->temporalLayerIndex, (int)pcs->sliceType, t_in, md_now());
            return;
        }
        if (!e->md_me && (svt_amd_host_alloc(lane, sizeof(SvtAmdMeLcuResult) * n, (void **)&e->md_me) ||
                          svt_amd_host_alloc(lane, sizeof(SvtAmdTmvpLcu) * n, (void **)&e->md_tmvp)))
            svt_hook_die("out of memory (mode-decision inter inputs)");
        SvtAmdMeLcuResult *me = e->md_me;
        SvtAmdTmvpLcu *tmvp = X.tmvp_enable ? e->md_tmvp : NULL;
        const EbReferenceObject_t *col =
            (const EbReferenceObject_t *)pcs->refPicPtrArray[pcs->sliceType == EB_B_PICTURE ? pcs->colocatedPuRefList : REF_LIST_0]->objectPtr;
        for (size_t i = 0; i < n; i++) {
            svt_md_fill_me(&me[i], pcs->ParentPcsPtr, (EB_U32)i);
            if (tmvp)
                svt_md_fill_tmvp(&tmvp[i], &col->tmvpMap[i]);
        }
        const double t_c1 = md_now();
        if (e->wide ? svt_amd_md_encode_picture_inter16(lane, e->pic, &P, &X, lcus, e->md_src16[0], W2, e->md_src16[1], e->md_src16[2], W2 / 2, ois, 0, me, 0, tmvp, e->md_out,
                                                        (SvtAmdLcuWork16 *)e->md_works, (SvtAmdLcuResult16 *)e->md_res)
                    : svt_amd_md_encode_picture_inter(lane, e->pic, &P, &X, lcus, sy, in->strideY, scb, scr, in->strideCb, ois, 0, me, 0, tmvp, e->md_out,
                                                      (SvtAmdLcuWork *)e->md_works, (SvtAmdLcuResult *)e->md_res))
            svt_hook_die("svt_amd_md_encode_picture_inter");
        t_dev = md_now() - t_c1;
        __atomic_add_fetch(&g_md_inter_pictures, 1, __ATOMIC_RELAXED);
    }
    e->md_ok = 1;
    {
        /* the call has returned: no kernel reads this picture's reference pictures any more (its LCUs are served from the records), so their cache slots are free to be
         * recycled NOW - not when this PictureControlSet_t object meets its next picture (idle objects of a large pool kept up to two references pinned each and
         * the cache grew instead of evicting) */
        const int pins[2] = {e->ref_pins_plus1[0] - 1, e->ref_pins_plus1[1] - 1};
        svt_hook_release_references(pins);
        e->ref_pins_plus1[0] = e->ref_pins_plus1[1] = 0;
    }
    if (!inter)
        t_dev = md_now() - t_c0;
    svt_hook_timeline("md_fill", pcs->pictureNumber, pcs->temporalLayerIndex, (int)pcs->sliceType, t_in, t_p0);
    svt_hook_timeline("md_refs", pcs->pictureNumber, pcs->temporalLayerIndex, (int)pcs->sliceType, t_p0, t_c0);
    svt_hook_timeline("md_call", pcs->pictureNumber, pcs->temporalLayerIndex, (int)pcs->sliceType, md_now() - t_dev, md_now());
    g_md_t_call += t_dev, g_md_t_fill += md_now() - t_in - t_dev;
    __atomic_add_fetch(&g_md_pictures, 1, __ATOMIC_RELAXED);
}

EB_ERRORTYPE __wrap_ModeDecisionLcu(SequenceControlSet_t *scs, PictureControlSet_t *pcs, const MdcLcuData_t *const mdcResultTbPtr,
                                    LargestCodingUnit_t *lcuPtr, EB_U16 lcuOriginX, EB_U16 lcuOriginY, EB_U32 lcuAddr, ModeDecisionContext_t *contextPtr)
{
    if (g_md_state == 0) {
        g_md_verify = getenv("SVT_HOOK_MD_VERIFY") != NULL;
        g_md_state = svt_hook_cfg("SVT_HOOK_MD") ? 1 : -1;
        g_md_skip_intra = svt_hook_cfg("SVT_HOOK_MD") && !strcmp(svt_hook_cfg("SVT_HOOK_MD"), "pb"); /* SVT_HOOK_MD=pb: only P / B pictures go to the device */
    }
    if (g_md_state < 0 || svt_hook_failed() || pcs->colorFormat != EB_YUV420 || (scs->lumaWidth & 7) || (scs->lumaHeight & 7))
        return __real_ModeDecisionLcu(scs, pcs, mdcResultTbPtr, lcuPtr, lcuOriginX, lcuOriginY, lcuAddr, contextPtr);
    svt_hook_note_callback(scs);
    const int wide = scs->staticConfig.encoderBitDepth > EB_8BIT; /* = EncDecContext_t.is16bit (EbEncDecProcess.c): the picture object holds 16-bit samples */
    EpPictureEntry *e = entry_lookup(scs, pcs, wide);
    int ok = 0;
    if (e && __atomic_load_n(&e->md_done_plus1, __ATOMIC_ACQUIRE) == pcs->pictureNumber + 1) { /* the picture's device call has returned: no lock, no lane */
        ok = e->md_ok;
        goto decided;
    }
    if (e && ep_lanes() < EP_LANES) { /* lanes are capped (SVT_HOOK_EP_LANES) and the object is warmed up: whether the picture takes the device call at all is settled without a lane */
        svt_hook_lock(&e->lock);
        if (e->md_picture_plus1 != pcs->pictureNumber + 1) {
            SvtAmdMdPicture P;
            SvtAmdMdInter X;
            const double t_in = md_now();
            if (!md_picture_level_ok(scs, pcs, contextPtr, &P, &X)) {
                e->md_picture_plus1 = pcs->pictureNumber + 1, e->md_ok = 0;
                __atomic_add_fetch(&g_md_left_pictures, 1, __ATOMIC_RELAXED);
                svt_hook_timeline("md_host", pcs->pictureNumber, pcs->temporalLayerIndex, (int)pcs->sliceType, t_in, md_now());
                __atomic_store_n(&e->md_done_plus1, pcs->pictureNumber + 1, __ATOMIC_RELEASE);
            }
        }
        const int settled = e->md_picture_plus1 == pcs->pictureNumber + 1 && __atomic_load_n(&e->md_done_plus1, __ATOMIC_ACQUIRE) == pcs->pictureNumber + 1;
        if (settled)
            ok = e->md_ok;
        svt_hook_unlock(&e->lock);
        if (settled)
            goto decided;
    }
    SvtAmdContext *root = svt_hook_device((uint16_t)scs->lumaWidth, (uint16_t)scs->lumaHeight);
    SvtAmdContext *lane = lane_claim(root);
    e = picture_entry(lane, scs, pcs, wide, 0);
    svt_hook_lock(&e->lock);
    if (e->md_picture_plus1 != pcs->pictureNumber + 1)
        md_picture(lane, e, scs, pcs, contextPtr);
    ok = e->md_ok;
    __atomic_store_n(&e->md_done_plus1, pcs->pictureNumber + 1, __ATOMIC_RELEASE);
    svt_hook_unlock(&e->lock);
    lane_release(lane);
decided:;
    if (!ok)
        return __real_ModeDecisionLcu(scs, pcs, mdcResultTbPtr, lcuPtr, lcuOriginX, lcuOriginY, lcuAddr, contextPtr);
    const SvtAmdMdLcuOut *o = &e->md_out[lcuAddr];
    if (g_md_verify) {
        const EB_ERRORTYPE rc = __real_ModeDecisionLcu(scs, pcs, mdcResultTbPtr, lcuPtr, lcuOriginX, lcuOriginY, lcuAddr, contextPtr);
        int bad = 0;
        for (EB_U32 i = 0; i < CU_MAX_COUNT;) { /* the final tree, as EncodePass walks it */
            const CodingUnit_t *cu = lcuPtr->codedLeafArrayPtr[i];
            if (cu->splitFlag != o->split[i]) {
                bad++;
                break;
            }
            if (cu->splitFlag) {
                i++;
                continue;
            }
            bad += cu->predictionModeFlag != o->pred_mode[i] || cu->predictionUnitArray->intraLumaMode != o->intra_luma_mode[i];
            if (cu->predictionModeFlag == INTER_MODE) {
                const PredictionUnit_t *pu = cu->predictionUnitArray;
                bad += pu->interPredDirectionIndex != o->inter_dir[i] || pu->mergeFlag != o->merge_flag[i] || pu->mv[0].x != o->mv[i][0][0] ||
                       pu->mv[0].y != o->mv[i][0][1] || pu->mv[1].x != o->mv[i][1][0] || pu->mv[1].y != o->mv[i][1][1];
            }
            i += DepthOffset[GetCodedUnitStats(i)->depth];
        }
        __atomic_add_fetch(&g_md_verified, 1, __ATOMIC_RELAXED);
        if (bad) {
            __atomic_add_fetch(&g_md_mismatch, 1, __ATOMIC_RELAXED);
            fprintf(stderr, "svt_hook_encdec: MD VERIFY picture %llu lcu %u: the device's tree differs from the reference's\n",
                    (unsigned long long)pcs->pictureNumber, lcuAddr);
            for (EB_U32 i = 0, shown = 0; i < CU_MAX_COUNT && shown < 6; i++) {
                const CodingUnit_t *cu = lcuPtr->codedLeafArrayPtr[i];
                if (cu->splitFlag != o->split[i] || (contextPtr->mdLocalCuUnit[i].testedCuFlag && (cu->predictionModeFlag != o->pred_mode[i] ||
                    cu->predictionUnitArray->intraLumaMode != o->intra_luma_mode[i] || contextPtr->mdLocalCuUnit[i].cost != o->cost[i]))) {
                    fprintf(stderr, "    leaf %u: split ref %d dev %d, tested ref %d dev %d, mode ref %d/%d dev %d/%d, cost ref %llu dev %llu\n", i, (int)cu->splitFlag,
                            o->split[i], (int)contextPtr->mdLocalCuUnit[i].testedCuFlag, o->tested[i], (int)cu->predictionModeFlag,
                            (int)cu->predictionUnitArray->intraLumaMode, o->pred_mode[i], o->intra_luma_mode[i],
                            (unsigned long long)contextPtr->mdLocalCuUnit[i].cost, (unsigned long long)o->cost[i]);
                    shown++;
                }
            }
        }
        return rc;
    }
    md_apply(lcuPtr, o, contextPtr->qp, contextPtr);
    __atomic_add_fetch(&g_md_lcus, 1, __ATOMIC_RELAXED);
    return EB_ErrorNone;
}

void svt_hook_encdec_report(FILE *out)
{
    if (g_md_state > 0)
        fprintf(out, "svt_hook_me: mode decision: %lu pictures (%lu of them P / B; %lu LCUs) decided AND encoded by ONE device call each (ModeDecisionLcu + EncodePass, "
                     "no per-candidate call); %lu pictures outside the device call left to the reference code; verification: %lu LCUs compared, %lu differ\n",
                g_md_pictures, g_md_inter_pictures, g_md_lcus, g_md_left_pictures, g_md_verified, g_md_mismatch);
    if (g_md_state > 0 && g_md_pictures)
        fprintf(out, "svt_hook_me: mode decision: per device-decided picture %.1f ms inside the device call (uploads, kernel, downloads), %.1f ms of host work around it "
                     "(controls, open-loop intra / motion-estimation / motion-field records, reference pictures: %.1f ms of it making the reference pictures and rate "
                     "tables resident)\n", 1e3 * g_md_t_call / g_md_pictures, 1e3 * g_md_t_fill / g_md_pictures, 1e3 * g_md_t_prepare / g_md_pictures);
    if (g_prep_n)
        fprintf(out, "svt_hook_me: picture objects prepared %lu times: %.2f ms reference pictures resident, %.2f ms rate tables + references set, %.2f ms begin (means)\n", g_prep_n,
                1e3 * g_prep_t[0] / g_prep_n, 1e3 * g_prep_t[1] / g_prep_n, 1e3 * g_prep_t[2] / g_prep_n);
    if (g_ep_state <= 0)
        return;
    fprintf(out, "svt_hook_me: encode pass: %lu LCUs encoded on the GPU (one call each; %lu of them with inter units, %lu inter units); left to the "
                 "reference code: %lu LCUs with units outside the device call, %lu under tools outside it, %lu in another sample format; %lu host LCU "
                 "borders handed over in %lu calls\n",
            g_ep_gpu, g_ep_inter_lcus, g_ep_inter_units, g_ep_cpu_units, g_ep_cpu_tools, g_ep_cpu_format, g_ep_borders, g_ep_puts);
    if (g_ep_refs)
        fprintf(out, "svt_hook_me: encode pass: %lu reference pictures finished on the device, %lu left to the upload path (an LCU outside the device "
                     "call, per-LCU SAO lambdas, or SAO without deblocking)\n", g_ep_refs_done, g_ep_refs_skipped);
    if (g_ep_verify)
        fprintf(out, "svt_hook_me: encode pass verification: %lu device-encoded LCUs compared with the reference's own EncodePass, %lu differ\n", g_ep_verified,
                g_ep_mismatch);
}
