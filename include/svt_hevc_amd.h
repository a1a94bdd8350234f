/*
 * svt_hevc_amd.h - C-ABI of the MI355X-native block-analysis hot path of SVT-HEVC.
 *
 * Two layers:
 *
 *  (1) LEAF layer ("svt_amd_<ReferenceName>"): one entry point per slot of the
 *      reference's [EB_ASM_TYPE_TOTAL] function-pointer tables, with the exact
 *      argument list of the reference typedef it replaces, on HOST pointers,
 *      synchronous.  A maintainer drops these into a third table slot
 *      (INTEGRATION.md).  They exist for per-call differential parity, not speed.
 *
 *  (2) BATCHED layer: one call per picture for the open-loop front half
 *      (pad + decimate + HME L0/L1/L2 + full-pel 85-PU search + AVC-style
 *      half/quarter-pel refinement + bi-pred + candidate sort), called from
 *      MotionEstimationKernel in place of its LCU loop
 *      (reference: Source/Lib/Codec/EbMotionEstimationProcess.c:706-820).
 *      Device state lives behind an opaque handle.
 *
 * Plain C types only.  All functions return 0 (SVT_AMD_OK) or a negative
 * SvtAmdStatus; nothing here falls back to a CPU implementation - if the HIP
 * device or code object is unavailable the call fails with SVT_AMD_ERR_DEVICE.
 */
#ifndef SVT_HEVC_AMD_H
#define SVT_HEVC_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVT_AMD_API __attribute__((visibility("default")))

typedef enum SvtAmdStatus {
    SVT_AMD_OK = 0,
    SVT_AMD_ERR_BAD_PARAM = -1,   /* maps to EB_ErrorBadParameter            (EbApi.h:114-129) */
    SVT_AMD_ERR_RESOURCES = -2,   /* maps to EB_ErrorInsufficientResources                     */
    SVT_AMD_ERR_DEVICE = -3       /* HIP failure; maps to EB_ErrorUndefined                    */
} SvtAmdStatus;

/* ------------------------------------------------------------------------- */
/* Shared data model                                                         */
/* ------------------------------------------------------------------------- */

#define SVT_AMD_LCU_SIZE 64
#define SVT_AMD_ME_PU_COUNT 85        /* MAX_ME_PU_COUNT, EbMotionEstimationLcuResults.h:14 */
#define SVT_AMD_PAD_FULL 68           /* lcuSize + ME_FILTER_TAP, EbEncHandle.c:901-904     */
#define SVT_AMD_PAD_QUARTER 32        /* EbEncHandle.c:911-914                              */
#define SVT_AMD_PAD_SIXTEENTH 16      /* EbEncHandle.c:921-924                              */

/* fractionalSearchMethod values, EbDefinitions.h:756-758 */
#define SVT_AMD_SUB_SAD_SEARCH 0
#define SVT_AMD_FULL_SAD_SEARCH 1
#define SVT_AMD_SSD_SEARCH 2

/* prediction direction, EbDefinitions.h (UNI_PRED_LIST_0/1, BI_PRED) */
#define SVT_AMD_UNI_PRED_LIST_0 0
#define SVT_AMD_UNI_PRED_LIST_1 1
#define SVT_AMD_BI_PRED 2

/*
 * Per-picture ME/HME controls.  Every field is a value the reference derives
 * on the host (out of the hot-path scope) and hands to MotionEstimateLcu:
 *   MeContext_t fields set by SetMeHmeParamsOq / SignalDerivationMeKernelOq
 *   (EbMotionEstimationProcess.c:54-120, 310-421) and PictureParentControlSet_t
 *   fields set by PictureDecision (EbPictureDecisionProcess.c:334-470).
 */
typedef struct SvtAmdMeParams {
    uint16_t luma_width;            /* SequenceControlSet_t.lumaWidth  (multiple of 8) */
    uint16_t luma_height;           /* SequenceControlSet_t.lumaHeight                  */
    uint8_t  num_lists;             /* 1: P picture (list 0 only), 2: B picture         */
    uint8_t  temporal_layer_index;  /* pcs->temporalLayerIndex                          */
    uint8_t  ref_pocs_equal;        /* refPicPocArray[0] == refPicPocArray[1]           */
    uint8_t  enable_hme_flag;       /* pcs->enableHmeFlag                               */
    uint8_t  enable_hme_level0;     /* pcs->enableHmeLevel0Flag                         */
    uint8_t  enable_hme_level1;
    uint8_t  enable_hme_level2;
    uint8_t  one_quadrant_hme;      /* MeContext_t.oneQuadrantHME                       */
    uint8_t  update_hme_search_center; /* MeContext_t.updateHmeSearchCenter             */
    uint8_t  num_hme_regions_w;     /* numberHmeSearchRegionInWidth  (<= 2)             */
    uint8_t  num_hme_regions_h;     /* numberHmeSearchRegionInHeight (<= 2)             */
    uint8_t  search_area_width;     /* MeContext_t.searchAreaWidth  (full-pel)          */
    uint8_t  search_area_height;
    uint8_t  fractional_search_method; /* SVT_AMD_*_SEARCH                              */
    uint8_t  fractional_search_model;  /* 0: all, 1: SuPelEnable gated, 2: off          */
    uint8_t  fractional_search_64x64;
    uint8_t  cu8x8_mode;            /* pcs->cu8x8Mode  (1 = no 8x8 refinement/bipred)   */
    uint8_t  cu16x16_mode;          /* pcs->cu16x16Mode                                 */
    uint16_t hme_l0_total_w;        /* hmeLevel0TotalSearchAreaWidth                    */
    uint16_t hme_l0_total_h;
    uint16_t hme_l0_w[2];           /* hmeLevel0SearchAreaInWidthArray                  */
    uint16_t hme_l0_h[2];
    uint16_t hme_l1_w[2];
    uint16_t hme_l1_h[2];
    uint16_t hme_l2_w[2];
    uint16_t hme_l2_h[2];
    uint16_t hme_l0_mult_x;         /* HME_LEVEL_0_SEARCH_AREA_MULTIPLIER_X[hl][tl], EbDefinitions.h:1322 */
    uint16_t hme_l0_mult_y;
    uint32_t lambda;                /* MeContext_t.lambda (SAD lambda, EbMotionEstimationProcess.c:690-698) */
    uint32_t mvd_bits[12];          /* MeContext_t.mvdBitsArray (NUMBER_OF_MVD_CASES)   */
} SvtAmdMeParams;

/* One ME candidate record per PU; same content as MeCuResults_t
 * (EbMotionEstimationLcuResults.h:58-74) with explicit widths. */
typedef struct SvtAmdMeCuResult {
    int16_t  x_mv_l0, y_mv_l0, x_mv_l1, y_mv_l1;   /* quarter-pel units */
    uint32_t distortion[3];                          /* sorted candidates */
    uint8_t  direction[3];
    uint8_t  total_me_candidate_index;
} SvtAmdMeCuResult;

/* Full per-LCU output of the front half (what MotionEstimateLcu leaves behind). */
typedef struct SvtAmdMeLcuResult {
    SvtAmdMeCuResult pu[SVT_AMD_ME_PU_COUNT];        /* raster-within-tier order = meResults[lcu][pu] */
    uint32_t best_sad[2][SVT_AMD_ME_PU_COUNT];       /* pLcuBestSad[list][0][], internal Z order      */
    uint32_t best_mv[2][SVT_AMD_ME_PU_COUNT];        /* pLcuBestMV, (y<<16)|x quarter-pel             */
    int16_t  hme_center_x[2], hme_center_y[2];       /* search centre fed to the full-pel search      */
    int16_t  search_origin_x[2], search_origin_y[2]; /* x/ySearchAreaOrigin[list][0]                  */
    uint8_t  search_w[2], search_h[2];               /* clipped search area                           */
} SvtAmdMeLcuResult;

/* ------------------------------------------------------------------------- */
/* BATCHED layer                                                             */
/* ------------------------------------------------------------------------- */

typedef struct SvtAmdContext SvtAmdContext;          /* opaque; owns device, stream, arenas */

/* Library/device bring-up.  Replaces nothing in the reference: called from
 * EbInitEncoder/EbDeinitEncoder (EbEncHandle.c:689,1519) by the hook overlay. */
SVT_AMD_API int svt_amd_context_create(int device_ordinal, uint16_t max_luma_width,
                                       uint16_t max_luma_height, int num_picture_slots,
                                       SvtAmdContext **out_ctx);
SVT_AMD_API void svt_amd_context_destroy(SvtAmdContext *ctx);
SVT_AMD_API const char *svt_amd_version(void);
/* OPT-IN process setting for hosts that keep several pictures in flight (lanes = HIP streams): asks the HIP runtime for 24 hardware queues
 * (GPU_MAX_HW_QUEUES, unless the user set it) so that every lane's stream has a queue of its own.  Only has an effect BEFORE the process's first
 * HIP call; the library never changes the environment on its own. */
SVT_AMD_API int svt_amd_runtime_env_defaults(void);
/* OPT-IN device setting for hosts whose threads wait on the device while OTHER host threads have work to do (the encoder binding: up to a dozen EncDec threads
 * wait 60 - 100 ms each for their picture's mode-decision kernel while the base-layer pictures are decided on the same 32 logical processors): the threads sleep
 * in the HIP runtime's waits (hipDeviceScheduleBlockingSync) instead of spinning on a completion signal, which costs every wait a wake-up (tens of microseconds)
 * and gives the cores back.  blocking = 0 restores the runtime's default.  A setting of the DEVICE, i.e. of every context of this process on it; best made before
 * the device's first context (svt_amd_context_create). */
SVT_AMD_API int svt_amd_host_wait_mode(int device_ordinal, int blocking);
SVT_AMD_API const char *svt_amd_last_error(void);

/*
 * Upload one source luma plane into picture slot `slot` and build, on the
 * device, everything the PA reference object holds for ME:
 *   padded full-resolution plane      (GeneratePadding,   EbMcp.c:1017)
 *   1/4 and 1/16 point-decimated planes + their padding
 *                                     (DecimateInputPicture, EbPictureAnalysisProcess.c:4139;
 *                                      Decimation2D :173)
 *   AVC-style half-pel planes b, h, j (EbHevcInterpolateSearchRegionAVC, EbMotionEstimation.c:645;
 *                                      AvcStyleLumaInterpolationFilterHorizontal/Vertical,
 *                                      C_DEFAULT/EbAvcStyleMcp_C.c:34,62), computed once per picture
 *                                      instead of once per LCU per list.
 * `luma` is a HOST pointer to width x height samples with `stride` bytes per row.
 */
SVT_AMD_API int svt_amd_picture_upload(SvtAmdContext *ctx, int slot, const uint8_t *luma,
                                       uint32_t stride, uint16_t width, uint16_t height);

/* Same, but the plane already lives in device memory (bench / multi-GPU path). */
SVT_AMD_API int svt_amd_picture_upload_device(SvtAmdContext *ctx, int slot, const void *d_luma,
                                              uint32_t stride, uint16_t width, uint16_t height);
/* Up to 256 pictures of the same geometry in ONE launch (slots[i] <- d_luma[i]). */
SVT_AMD_API int svt_amd_picture_upload_device_batch(SvtAmdContext *ctx, int num, const int *slots,
                                                    const void *const *d_luma, uint32_t stride,
                                                    uint16_t width, uint16_t height);

/*
 * Motion estimation of one whole picture (all LCUs), replaces the LCU loop of
 * MotionEstimationKernel (EbMotionEstimationProcess.c:706-820) and
 * MotionEstimateLcu (EbMotionEstimation.c:3671-4450).
 *   cur_slot          picture being coded
 *   ref_slot[l]       PA reference picture of list l (ref_slot[1] ignored for P)
 *   out               HOST array of picWidthInLcu*picHeightInLcu records, raster LCU order
 * Blocking: returns when `out` is filled.
 */
SVT_AMD_API int svt_amd_me_picture(SvtAmdContext *ctx, const SvtAmdMeParams *params,
                                   int cur_slot, const int ref_slot[2],
                                   SvtAmdMeLcuResult *out);

/* Asynchronous form: results stay in device memory (slot-owned buffer) until
 * svt_amd_me_picture_fetch copies them out.  Used by bench.py and by a host
 * that overlaps ME of picture n+1 with EncDec of picture n. */
SVT_AMD_API int svt_amd_me_picture_launch(SvtAmdContext *ctx, const SvtAmdMeParams *params,
                                          int cur_slot, const int ref_slot[2]);
SVT_AMD_API int svt_amd_me_picture_fetch(SvtAmdContext *ctx, int cur_slot, SvtAmdMeLcuResult *out);
/*
 * Batched form: motion estimation of up to 256 pictures in ONE launch (grid = pictures x LCUs).
 * The front half is open loop, so every picture whose source and reference pictures have been
 * uploaded can be searched at once; this is what fills the 256 CUs (a single 1080p picture is
 * only 510 workgroups).  Results land in each current slot's result buffer (svt_amd_me_picture_fetch).
 * Reference analogue: several pictures are in flight across the 60 ME segment tasks per picture
 * (EbEncHandle.c:1680-1681, pictureControlSetPoolInitCount :1796-1809).
 */
typedef struct SvtAmdMeJob {
    SvtAmdMeParams params;
    int32_t cur_slot;
    int32_t ref_slot[2];
} SvtAmdMeJob;
SVT_AMD_API int svt_amd_me_batch_launch(SvtAmdContext *ctx, const SvtAmdMeJob *jobs, int num_jobs);

/* LCU-range form for multi-GPU sharding of one picture by LCU rows: only
 * records [lcu_begin, lcu_end) of the slot's result buffer are written.  The reference's
 * analogue is the 6x10 ME segment grid (EbEncHandle.c:1680-1681). */
SVT_AMD_API int svt_amd_me_picture_range_launch(SvtAmdContext *ctx, const SvtAmdMeParams *params,
                                                int cur_slot, const int ref_slot[2],
                                                uint32_t lcu_begin, uint32_t lcu_end);
SVT_AMD_API int svt_amd_synchronize(SvtAmdContext *ctx);
/* plain device memory on the context's GPU for C hosts that keep reference pictures / planes resident (blocking copies) */
SVT_AMD_API int svt_amd_device_alloc(SvtAmdContext *ctx, size_t bytes, void **d_ptr);
SVT_AMD_API int svt_amd_device_free(SvtAmdContext *ctx, void *d_ptr);
SVT_AMD_API int svt_amd_device_copy(SvtAmdContext *ctx, void *d_dst, const void *d_src, size_t bytes); /* device to device, blocking */
SVT_AMD_API int svt_amd_device_upload(SvtAmdContext *ctx, void *d_dst, const void *src, size_t bytes);
SVT_AMD_API int svt_amd_device_download(SvtAmdContext *ctx, void *dst, const void *d_src, size_t bytes);

/*
 * Open-loop intra search (OIS) of one whole picture: replaces the second LCU loop of
 * MotionEstimationKernel (EbMotionEstimationProcess.c:786-797) and OpenLoopIntraSearchLcu
 * (EbMotionEstimation.c:5053-5320): per 32x32 / 16x16 / 8x8 CU of every LCU, predict from SOURCE
 * neighbours (UpdateNeighborSamplesArrayOpenLoop, EbIntraPrediction.c:5222; out-of-picture = 128, no
 * smoothing) for the mode subset the slice type / OIS point selects, NxM SAD, candidate injection.
 */
typedef struct SvtAmdOisParams {
    uint16_t luma_width, luma_height;          /* SequenceControlSet_t.lumaWidth/Height                   */
    uint8_t  slice_is_intra;                   /* pcs->sliceType == EB_I_PICTURE                          */
    uint8_t  temporal_layer_index;             /* pcs->temporalLayerIndex                                 */
    uint8_t  limit_ois_to_dc_mode;             /* pcs->limitOisToDcModeFlag (EbPictureDecisionProcess.c:417) */
    uint8_t  skip_ois_8x8;                     /* pcs->skipOis8x8 (:455)                                  */
    uint8_t  cu8x8_mode;                       /* pcs->cu8x8Mode                                          */
    uint8_t  ois_kernel_level;                 /* MotionEstimationContext_t.oisKernelLevel (EbMotionEstimationProcess.c:356) */
    uint8_t  ois_th_set;                       /* .oisThSet (:371-393)                                    */
    uint8_t  set_best_ois_distortion_to_valid; /* .setBestOisDistortionToValid (:396)                     */
} SvtAmdOisParams;

#define SVT_AMD_OIS_MAX_CAND 18                /* MAX_OIS_2, EbCodingUnit.h:61 */
/* candidate word = OisCandidate_t.oisResults (EbCodingUnit.h:91-101): bits 0-19 distortion, 20 validDistortion,
 * 24-31 intraMode.  The reference only touches some bitfields of some entries per call (the rest keeps whatever
 * the PCS pool held); the three bits the reference leaves unnamed say which: */
#define SVT_AMD_OIS_W_DIST  (1u << 21)         /* this call wrote .distortion      */
#define SVT_AMD_OIS_W_VALID (1u << 22)         /* this call wrote .validDistortion */
#define SVT_AMD_OIS_W_MODE  (1u << 23)         /* this call wrote .intraMode       */
typedef struct SvtAmdOisLcuResult {
    uint32_t candidate[SVT_AMD_ME_PU_COUNT][SVT_AMD_OIS_MAX_CAND]; /* by rasterScanCuIndex; [0] (64x64) unused  */
    uint8_t  total_intra_luma_mode[SVT_AMD_ME_PU_COUNT];           /* 0xFF = left untouched by this call        */
    uint8_t  pad[3];
} SvtAmdOisLcuResult;
/* me: HOST array of the picture's ME results (distortion[0] of PUs 1..84 is read) or NULL - then the results of the
 * last ME launch for cur_slot, still resident on the device, are used (NULL is also right for I pictures and
 * limit_ois_to_dc_mode).  out: HOST array, one record per LCU, raster order.  Blocking. */
SVT_AMD_API int svt_amd_ois_picture(SvtAmdContext *ctx, const SvtAmdOisParams *params, int cur_slot,
                                    const SvtAmdMeLcuResult *me, SvtAmdOisLcuResult *out);
SVT_AMD_API int svt_amd_ois_picture_launch(SvtAmdContext *ctx, const SvtAmdOisParams *params, int cur_slot);
SVT_AMD_API int svt_amd_ois_picture_fetch(SvtAmdContext *ctx, int cur_slot, SvtAmdOisLcuResult *out);
/*
 * Collocated zero-motion SAD at 1/16 resolution and the background classes derived from it: replaces
 * ComputeDecimatedZzSad (EbMotionEstimationProcess.c:176-300, called per ME segment at :828 when lookAheadDistance != 0
 * and pictureNumber > 0).  prev_slot = the previous picture in display order; results are what the reference stores
 * into the PREVIOUS picture's zzCostArray / nonMovingIndexArray.  out: HOST array, one record per LCU.  Blocking.
 */
typedef struct SvtAmdZzLcu {
    uint32_t sad;              /* decimatedLcuCollocatedSad (0xFFFFFFFF for incomplete LCUs)     */
    uint8_t  zz_cost;          /* zzCostArray[lcu]: 0 / 3 / 10 / 20 / 30, 0xFF = INVALID_ZZ_COST */
    uint8_t  non_moving_index; /* nonMovingIndexArray[lcu]: 0 / 10 / 20 / 30                     */
    uint8_t  pad[2];
} SvtAmdZzLcu;
SVT_AMD_API int svt_amd_zz_sad_picture(SvtAmdContext *ctx, int cur_slot, int prev_slot, SvtAmdZzLcu *out);

/*
 * Front-end pipeline (the asynchronous product path of the batched boundary `hip_me_picture`): replaces, per
 * picture, the LCU loops of MotionEstimationKernel (Codec/EbMotionEstimationProcess.c:706-820) with
 *     upload (pinned staging, asynchronous H2D) -> plane building -> ME of both lists -> OIS -> results into pinned host
 *     memory (asynchronous D2H)
 * queued on ONE stream ("lane") and overlapped with the other lanes' copies and kernels.  A lane is a context forked from
 * the one that owns the picture slots (svt_amd_context_fork): all lanes see the same slots, each has its own stream,
 * descriptors, timers and pinned result buffers.  The owning context is itself a lane.  Calls on one lane are serialised by
 * the caller; different lanes may be driven from different threads.
 *   svt_amd_picture_upload_async  copies `luma` into pinned staging before returning (caller may reuse it), queues the rest
 *   svt_amd_frontend_submit       queues ME (has_me) and/or OIS (has_ois) of cur_slot + the result copies; never blocks
 *   svt_amd_frontend_wait         blocks until the lane's job is complete; *me / *ois point at the lane's pinned buffers
 *                                 (one record per LCU, raster order), valid until svt_amd_frontend_release
 */
typedef struct SvtAmdFrontendJob {
    int32_t cur_slot;
    int32_t ref_slot[2];
    uint8_t has_me;            /* P / B pictures */
    uint8_t has_ois;
    uint8_t compact;           /* 1: the lane's pinned buffers receive the COMPACT records (below) instead of the full ones */
    uint8_t pad;
    SvtAmdMeParams me;
    SvtAmdOisParams ois;
} SvtAmdFrontendJob;
SVT_AMD_API int svt_amd_context_fork(SvtAmdContext *parent, SvtAmdContext **out_lane);
SVT_AMD_API int svt_amd_picture_upload_async(SvtAmdContext *lane, int slot, const uint8_t *luma, uint32_t stride,
                                             uint16_t width, uint16_t height);
SVT_AMD_API int svt_amd_picture_publish(SvtAmdContext *lane, int slot);
SVT_AMD_API int svt_amd_frontend_submit(SvtAmdContext *lane, const SvtAmdFrontendJob *job);
SVT_AMD_API int svt_amd_frontend_wait(SvtAmdContext *lane, const SvtAmdMeLcuResult **me, const SvtAmdOisLcuResult **ois);
SVT_AMD_API int svt_amd_frontend_release(SvtAmdContext *lane);
/* Cross-lane ordering without host waits (for hosts that build their own copy-in -> compute -> copy-out pipelines): a lane records
 * its event `index` (0..7) at the current end of its stream; another lane's stream waits for the latest record of it. */
SVT_AMD_API int svt_amd_lane_event_record(SvtAmdContext *lane, int index);
SVT_AMD_API int svt_amd_lane_event_wait(SvtAmdContext *lane, SvtAmdContext *source, int index);
/* Compact wire format of the records (what the reference side of the boundary consumes; the full records also carry the internal
 * best-SAD / MV arrays and 18 candidate slots per CU for parity tests): per LCU
 *   ME :  SvtAmdMeCuResult[85]                                                        (2,040 B instead of 3,420)
 *   OIS:  uint32_t candidate[85][nc] + uint8_t total_intra_luma_mode[85] + 3 pad      (85 * nc * 4 + 88 B instead of 6,208)
 * nc = svt_amd_ois_compact_candidates(params) = the most candidates the picture's path writes per CU: 7 on I pictures, 9 on P / B
 * pictures, 18 with ois_kernel_level (MAX_OIS_0 / _1 / _2, Codec/EbCodingUnit.h:59-61).  The device packs, then copies. */
#define SVT_AMD_OIS_COMPACT_BYTES(nc) (SVT_AMD_ME_PU_COUNT * (nc) * 4 + 88)
SVT_AMD_API int svt_amd_ois_compact_candidates(const SvtAmdOisParams *params);
SVT_AMD_API int svt_amd_me_picture_fetch_compact_async(SvtAmdContext *ctx, int cur_slot, SvtAmdMeCuResult *out);
SVT_AMD_API int svt_amd_ois_picture_fetch_compact_async(SvtAmdContext *ctx, int cur_slot, int candidates, void *out);
/* n pictures at once (batched hosts): the slots' records packed into contiguous DEVICE arrays, picture i of the batch at index i of
 * d_me (LCUs x 85 records each) / d_ois (LCUs x SVT_AMD_OIS_COMPACT_BYTES(candidates) bytes each); either may be NULL.  Run it on
 * the lane that ran the searches and move the arrays with one copy each on any lane (svt_amd_device_download_async). */
SVT_AMD_API int svt_amd_records_pack_batch_async(SvtAmdContext *ctx, const int *slots, int n, int candidates, SvtAmdMeCuResult *d_me,
                                                 void *d_ois);
/* Start-up, outside any timed run: pins the staging buffers of every picture slot (root) and the lane's result buffers, and sends
 * one dummy picture through upload -> planes -> ME -> OIS so that the kernels' code objects are loaded.  Once per lane (and root). */
SVT_AMD_API int svt_amd_frontend_warmup(SvtAmdContext *ctx);
/* building blocks of the same pipeline for batched hosts (bench.py): stream-ordered copies that do not wait, pinned host
 * memory to copy from / into, and the per-slot result copies; svt_amd_synchronize(lane) completes them */
SVT_AMD_API int svt_amd_device_upload_async(SvtAmdContext *ctx, void *d_dst, const void *src, size_t bytes);
SVT_AMD_API int svt_amd_device_download_async(SvtAmdContext *ctx, void *dst, const void *d_src, size_t bytes);
SVT_AMD_API int svt_amd_host_alloc(SvtAmdContext *ctx, size_t bytes, void **h_ptr);
SVT_AMD_API int svt_amd_host_free(SvtAmdContext *ctx, void *h_ptr);
/* Page-locks a buffer the CALLER owns (hipHostRegister) so that copies from / into it go through the DMA engines instead of the runtime's staged copy of pageable
 * memory - which is a shader (blit) kernel on this runtime and competes for compute units with the persistent mode-decision kernels.  For host buffers that live as long
 * as the encoder (the reference's input-picture and reference-picture pools, Codec/EbEncHandle.c:1794-1813): a process-wide table remembers (pointer, size), a second
 * call for the same range costs a lookup.  Returns SVT_AMD_OK also when the runtime refuses the range (the copies then take the pageable path as before).
 * svt_amd_host_unregister_all() releases every registration (before the caller frees the buffers). */
SVT_AMD_API int svt_amd_host_register(SvtAmdContext *ctx, const void *h_ptr, size_t bytes);
SVT_AMD_API int svt_amd_host_unregister_all(SvtAmdContext *ctx);
SVT_AMD_API int svt_amd_me_picture_fetch_async(SvtAmdContext *ctx, int cur_slot, SvtAmdMeLcuResult *out);
SVT_AMD_API int svt_amd_ois_picture_fetch_async(SvtAmdContext *ctx, int cur_slot, SvtAmdOisLcuResult *out);

/*
 * Multi-GPU exchange of the closed-loop half (SURVEY 8e "EncDec with tiles").  Tiles are independent inside a picture
 * (Codec/EbEntropyCoding.c:6346-6351, EbEncDecProcess.c:2743-2760), so rank r encodes a rectangle of whole tiles; motion vectors
 * of later pictures may cross tile borders (EbEncHandle.c:2757), so every FINISHED reference picture is all-gathered once
 * (after DLF / SAO, before it is padded and referenced: EbEncDecProcess.c:1806) - the only data-path collective.
 *   svt_amd_tile_partition: the reference's uniform tile grid (tileColStartLcu[c] = c * widthInLcu / cols,
 *     Codec/EbPictureControlSet.c:743-750) -> rank_rect[world] (luma samples; every rank a rectangle of whole tiles) and,
 *     optionally, tile_rank[tile_rows * tile_cols].  Host code, no device needed.
 *   svt_amd_comm_*: RCCL communicator of a context (one process per GPU; rank 0 makes the id, the host passes it round -
 *     torch.distributed / MPI / a file).  librccl is opened at run time.
 *   svt_amd_recon_exchange: pack own rectangle of Y / Cb / Cr (4:2:0) -> ncclAllGather -> unpack the others into the local planes;
 *     stream-ordered on the context's stream.  d_planes: device pointers to sample (0,0); pitches in bytes.
 *   svt_amd_recon_pack: the local halves alone (planes <-> slot r of a caller-owned buffer), for hosts with their own transport.
 */
typedef struct SvtAmdRect { uint16_t x, y, w, h; } SvtAmdRect;
typedef struct SvtAmdCommId { char bytes[128]; } SvtAmdCommId;
SVT_AMD_API int svt_amd_tile_partition(uint16_t luma_width, uint16_t luma_height, int tile_cols, int tile_rows, int world,
                                       SvtAmdRect *rank_rect, int *tile_rank);
SVT_AMD_API int svt_amd_comm_unique_id(SvtAmdCommId *out);
SVT_AMD_API int svt_amd_comm_init(SvtAmdContext *ctx, int world, int rank, const SvtAmdCommId *id);
SVT_AMD_API int svt_amd_comm_destroy(SvtAmdContext *ctx);
SVT_AMD_API int svt_amd_recon_exchange(SvtAmdContext *ctx, void *const d_planes[3], const uint32_t pitch_bytes[3], int bytes_per_sample,
                                       const SvtAmdRect *rects, int world, int rank);
SVT_AMD_API int svt_amd_recon_pack(SvtAmdContext *ctx, void *const d_planes[3], const uint32_t pitch_bytes[3], int bytes_per_sample,
                                   const SvtAmdRect *rects, int world, int r, void *d_slots, size_t slot_bytes, int to_slot);

/*
 * Picture-analysis statistics: what GatheringPictureStatistics (Codec/EbPictureAnalysisProcess.c:3995) computes from the planes the
 * front half already holds in HBM, one call per picture slot (after svt_amd_picture_upload*):
 *   per LCU  variance[85] / y_mean[85] of the 64x64, 32x32, 16x16 and 8x8 blocks in the order of the motion-estimation units
 *            (ME_TIER_ZERO_PU_*: 64x64, four 32x32, sixteen 16x16, sixty-four 8x8, each raster) - ComputeBlockMeanComputeVariance (:1646): sums and
 *            sums of squares over the EVEN rows of each 8x8 block, means in 8 / 16 fractional bits averaged up the tree with >> 2;
 *   per picture  the luma histograms of the regions_w x regions_h regions of the 1/16 picture (bins start at 1 and end << 4,
 *            SubSampleLumaGeneratePixelIntensityHistogramBins :3384), the regions' average intensity and the picture's luma sum (<< 4 per region).
 * out: HOST, LCUs of the picture in raster order; histogram: HOST [regions_w][regions_h][256] or NULL; region_average: HOST [regions_w][regions_h]
 * or NULL; sum_luma: HOST or NULL.  Blocking.
 */
typedef struct SvtAmdPaLcuStats {
    uint16_t variance[SVT_AMD_ME_PU_COUNT];  /* pictureControlSetPtr->variance[lcu]  */
    uint8_t y_mean[SVT_AMD_ME_PU_COUNT];     /* pictureControlSetPtr->yMean[lcu]     */
    uint8_t pad;
} SvtAmdPaLcuStats;
SVT_AMD_API int svt_amd_picture_stats(SvtAmdContext *ctx, int slot, SvtAmdPaLcuStats *out, int regions_w, int regions_h, uint32_t *histogram,
                                      uint8_t *region_average, uint64_t *sum_luma);

/*
 * Source-based-operations input of the mode-decision configuration: CalculateAcEnergy (Codec/EbSourceBasedOperationsProcess.c:302-362) of
 * every LCU of the picture the slot holds - the AC energy (ComputeNxMSatdSadLCU, Codec/EbPictureOperators.c:232: 8x8 Hadamard sums minus the DC terms >> 2)
 * of the 64x64 and of its four 32x32, the values pictureControlSetPtr->lcuYSrcEnergyCuArray[lcu][0..4] carries into DeriveDefaultSegments
 * (EbModeDecisionConfiguration.c:409-426) and the encode pass's skin / contour tests (EbModeDecisionProcess.c:542).  LCUs the picture does not cover completely
 * get the reference's 100000000 (:351).  Which pictures the values are read for (I pictures outside low-delay P) is the caller's rule.
 * out: HOST, [LCUs in raster order][5].  Blocking.
 */
SVT_AMD_API int svt_amd_picture_ac_energy(SvtAmdContext *ctx, int slot, uint64_t *out);

/* Batched form (grid = pictures x LCUs), each job reading the ME results its slot holds on the device. */
typedef struct SvtAmdOisJob {
    SvtAmdOisParams params;
    int32_t cur_slot;
} SvtAmdOisJob;
SVT_AMD_API int svt_amd_ois_batch_launch(SvtAmdContext *ctx, const SvtAmdOisJob *jobs, int num_jobs);


/* Device-side timing of the launches issued between begin/end on the context's
 * own stream (HIP events); used for roofline.achieved in bench.py. */
SVT_AMD_API int svt_amd_timer_begin(SvtAmdContext *ctx);
SVT_AMD_API int svt_amd_timer_end(SvtAmdContext *ctx, float *elapsed_ms);
/* Average duration (ms) and launch count of the named kernel class since the
 * last svt_amd_timer_begin; kernel_class in {"me_search","me_subpel","prep"} */
SVT_AMD_API int svt_amd_kernel_time(SvtAmdContext *ctx, const char *kernel_class,
                                    float *avg_ms, int *launches);

/* Development aid: per-workgroup phase clock stamps of the next batched ME launch
 * (call with out == NULL to arm for `workgroups` = jobs x max LCUs, again with a buffer of
 * workgroups*16 u64 to collect).  tools/me_phase_profile.py prints the breakdown. */
SVT_AMD_API int svt_amd_debug_me_phase_profile(SvtAmdContext *ctx, size_t workgroups,
                                               unsigned long long *out);

/* Debug/parity access to the device-built planes of a slot (host copies).
 * which: 0 full, 1 quarter, 2 sixteenth, 3 half-pel b, 4 half-pel h, 5 half-pel j.
 * Copies the whole padded allocation; out_stride / out_pad describe it. */
SVT_AMD_API int svt_amd_picture_read_plane(SvtAmdContext *ctx, int slot, int which, uint8_t *dst,
                                           size_t dst_capacity, uint32_t *out_stride,
                                           uint32_t *out_pad, uint32_t *out_rows);

/* ------------------------------------------------------------------------- */
/* LEAF layer (table-slot replacements; host pointers; synchronous)          */
/* Each comment names the reference typedef/table and the C_DEFAULT symbol.  */
/* ------------------------------------------------------------------------- */

/* EB_SADKERNELNxM_TYPE, NxMSadKernel_funcPtrArray (EbComputeSAD.h:95);
 * C peer FastLoop_NxMSadKernel (C_DEFAULT/EbComputeSAD_C.c:147). */
SVT_AMD_API uint32_t svt_amd_NxMSadKernel(const uint8_t *src, uint32_t srcStride,
                                          const uint8_t *ref, uint32_t refStride,
                                          uint32_t height, uint32_t width);

/* EB_SADLOOPKERNELNxM_TYPE, NxMSadLoopKernel_funcPtrArray (EbComputeSAD.h:152);
 * C peer SadLoopKernel (C_DEFAULT/EbComputeSAD_C.c:170). */
SVT_AMD_API void svt_amd_SadLoopKernel(const uint8_t *src, uint32_t srcStride,
                                       const uint8_t *ref, uint32_t refStride,
                                       uint32_t height, uint32_t width, uint64_t *bestSad,
                                       int16_t *xSearchCenter, int16_t *ySearchCenter,
                                       uint32_t srcStrideRaw, int16_t searchAreaWidth,
                                       int16_t searchAreaHeight);

/* EB_SADAVGKERNELNxM_TYPE, NxMSadAveragingKernel_funcPtrArray (EbComputeSAD.h:123);
 * C peer CombinedAveragingSAD (C_DEFAULT/EbComputeSAD_C.c:14). */
SVT_AMD_API uint32_t svt_amd_NxMSadAveragingKernel(const uint8_t *src, uint32_t srcStride,
                                                   const uint8_t *ref1, uint32_t ref1Stride,
                                                   const uint8_t *ref2, uint32_t ref2Stride,
                                                   uint32_t height, uint32_t width);

/* GetEightHorizontalSearchPointResults_8x8_16x16_funcPtrArray (EbComputeSAD.h);
 * C peer GetEightHorizontalSearchPointResults_8x8_16x16_PU (EbComputeSAD_C.c:252). */
SVT_AMD_API void svt_amd_GetEightHorizontalSearchPointResults_8x8_16x16_PU(
    const uint8_t *src, uint32_t srcStride, const uint8_t *ref, uint32_t refStride,
    uint32_t *pBestSad8x8, uint32_t *pBestMV8x8, uint32_t *pBestSad16x16,
    uint32_t *pBestMV16x16, uint32_t mv, uint16_t *pSad16x16);

/* GetEightHorizontalSearchPointResults_32x32_64x64_funcPtrArray;
 * C peer GetEightHorizontalSearchPointResults_32x32_64x64 (EbComputeSAD_C.c:372). */
SVT_AMD_API void svt_amd_GetEightHorizontalSearchPointResults_32x32_64x64(
    const uint16_t *pSad16x16, uint32_t *pBestSad32x32, uint32_t *pBestSad64x64,
    uint32_t *pBestMV32x32, uint32_t *pBestMV64x64, uint32_t mv);

/* SadCalculation_8x8_16x16_funcPtrArray (EbMeSadCalculation.h:106);
 * C peer SadCalculation_8x8_16x16 (C_DEFAULT/EbMeSadCalculation_C.c:14). */
SVT_AMD_API void svt_amd_SadCalculation_8x8_16x16(
    const uint8_t *src, uint32_t srcStride, const uint8_t *ref, uint32_t refStride,
    uint32_t *pBestSad8x8, uint32_t *pBestSad16x16, uint32_t *pBestMV8x8,
    uint32_t *pBestMV16x16, uint32_t mv, uint32_t *pSad16x16);

/* SadCalculation_32x32_64x64_funcPtrArray; C peer EbMeSadCalculation_C.c:64. */
SVT_AMD_API void svt_amd_SadCalculation_32x32_64x64(
    const uint32_t *pSad16x16, uint32_t *pBestSad32x32, uint32_t *pBestSad64x64,
    uint32_t *pBestMV32x32, uint32_t *pBestMV64x64, uint32_t mv);

/* AvcStyleUniPredLumaIFFunctionPtrArray[..][1|2] (EbAvcStyleMcp.h:111);
 * C peers AvcStyleLumaInterpolationFilterHorizontal/Vertical (EbAvcStyleMcp_C.c:34,62). */
SVT_AMD_API void svt_amd_AvcStyleLumaInterpolationFilterHorizontal(
    const uint8_t *refPic, uint32_t srcStride, uint8_t *dst, uint32_t dstStride,
    uint32_t puWidth, uint32_t puHeight, uint8_t *tempBuf, uint32_t fracPos);
SVT_AMD_API void svt_amd_AvcStyleLumaInterpolationFilterVertical(
    const uint8_t *refPic, uint32_t srcStride, uint8_t *dst, uint32_t dstStride,
    uint32_t puWidth, uint32_t puHeight, uint8_t *tempBuf, uint32_t fracPos);

/* PictureAverageArray (EbAvcStyleMcp.h:129); C peer PictureAverageKernel
 * (C_DEFAULT/EbPictureOperators_C.c:46). */
SVT_AMD_API void svt_amd_PictureAverageKernel(const uint8_t *src0, uint32_t src0Stride,
                                              const uint8_t *src1, uint32_t src1Stride,
                                              uint8_t *dst, uint32_t dstStride,
                                              uint32_t areaWidth, uint32_t areaHeight);

/* SpatialFullDistortionKernel_funcPtrArray (EbPictureOperators.h:633);
 * C peer SpatialFullDistortionKernel (C_DEFAULT/EbPictureOperators_C.c:643). */
SVT_AMD_API uint64_t svt_amd_SpatialFullDistortionKernel(const uint8_t *input, uint32_t inputStride,
                                                         const uint8_t *recon, uint32_t reconStride,
                                                         uint32_t areaWidth, uint32_t areaHeight);

/* Decimation2D (EbPictureAnalysisProcess.c:173) - not table-dispatched in the
 * reference but the producer of the HME inputs. */
SVT_AMD_API void svt_amd_Decimation2D(const uint8_t *inputSamples, uint32_t inputStride,
                                      uint32_t inputAreaWidth, uint32_t inputAreaHeight,
                                      uint8_t *decimSamples, uint32_t decimStride,
                                      uint32_t decimStep);

/* ------------------------------------------------------------------------- */
/* EncDec leaf families: residual, transforms, quantisation, distortion, SATD */
/* ------------------------------------------------------------------------- */

/* BATCHED forms (device pointers; block b of a size x size batch at base + b*size*size).
 * kind: 0 DCT, 1 low-precision "Estimate" DCT (32/16 only), 2 DST (4x4 only).
 * Replace the per-TU calls EstimateTransform/EncodeTransform -> Transform* (EbTransforms.c:3268,3343;
 * tables EbTransforms.h:339-510), EstimateInvTransform/EncodeInvTransform (:3455,3502),
 * QuantizeInvQuantize via UnifiedQuantizeInvQuantize (EbTransforms.c:2978), PictureFullDistortionLuma
 * -> FullDistortionKernel*_32bit (EbPictureOperators.c:397; table EbPictureOperators.h:503) and
 * Compute8x8Satd/Compute4x4Satd (EbPictureOperators.c:186-262, EbHmCode.c:41). */
SVT_AMD_API int svt_amd_fwd_transform_batch(SvtAmdContext *ctx, int kind, int size, uint32_t bitIncrement,
                                            const int16_t *d_residual, int16_t *d_coeff, uint32_t nblocks);
SVT_AMD_API int svt_amd_inv_transform_batch(SvtAmdContext *ctx, int kind, int size, uint32_t bitIncrement,
                                            const int16_t *d_coeff, int16_t *d_residual, uint32_t nblocks);
/* The 32x32 transforms as integer-MFMA products (v_mfma_i32_32x32x32_i8; txfm_mfma.hip): same results as the two calls above for
 * size 32 (kind 0 DCT / 1 "Estimate" DCT forward; inverse DCT), every int16 input - blocks of the Estimate form whose 16-bit
 * butterfly levels would wrap (C_DEFAULT/EbTransforms_C.c:492-520; impossible for 8-bit residuals) are redone by the VALU kernel. */
SVT_AMD_API int svt_amd_fwd_transform_mfma_batch(SvtAmdContext *ctx, int kind, int size, uint32_t bitIncrement,
                                                 const int16_t *d_residual, int16_t *d_coeff, uint32_t nblocks);
SVT_AMD_API int svt_amd_inv_transform_mfma_batch(SvtAmdContext *ctx, int size, uint32_t bitIncrement, const int16_t *d_coeff,
                                                 int16_t *d_residual, uint32_t nblocks);
SVT_AMD_API int svt_amd_quantize_batch(SvtAmdContext *ctx, int size, uint32_t qFunc, uint32_t q_offset,
                                       int32_t shiftedQBits, int32_t shiftedFFunc, int32_t iq_offset,
                                       int32_t shiftNum, const int16_t *d_coeff, int16_t *d_quant,
                                       int16_t *d_recon, uint32_t *d_nz, uint32_t nblocks);
/* mode: 0 FullDistortionKernel_32bit, 1 ...CbfZero_32bit, 2 ...Intra_32bit; d_result = 2 x u64 per block */
SVT_AMD_API int svt_amd_full_distortion_batch(SvtAmdContext *ctx, int size, int mode, const int16_t *d_coeff,
                                              const int16_t *d_recon, uint64_t *d_result, uint32_t nblocks);
SVT_AMD_API int svt_amd_satd_batch(SvtAmdContext *ctx, int size, const int16_t *d_diff, uint64_t *d_satd,
                                   uint32_t nblocks);

/* LEAF forms: EB_TRANS_COEFF_* tables (EbTransforms.h:339-510); C peers C_DEFAULT/EbTransforms_C.c:1602-2119 */
#define SVT_AMD_DECL_TRANSFORM(name)                                                                      \
    SVT_AMD_API void svt_amd_##name(int16_t *src, const uint32_t srcStride, int16_t *dst,                 \
                                    const uint32_t dstStride, int16_t *transformInnerArrayPtr,            \
                                    uint32_t bitIncrement);
SVT_AMD_DECL_TRANSFORM(Transform32x32)
SVT_AMD_DECL_TRANSFORM(Transform32x32Estimate)
SVT_AMD_DECL_TRANSFORM(Transform16x16)
SVT_AMD_DECL_TRANSFORM(Transform16x16Estimate)
SVT_AMD_DECL_TRANSFORM(Transform8x8)
SVT_AMD_DECL_TRANSFORM(Transform4x4)
SVT_AMD_DECL_TRANSFORM(DstTransform4x4)
SVT_AMD_DECL_TRANSFORM(InvTransform32x32)
SVT_AMD_DECL_TRANSFORM(InvTransform16x16)
SVT_AMD_DECL_TRANSFORM(InvTransform8x8)
SVT_AMD_DECL_TRANSFORM(InvTransform4x4)
SVT_AMD_DECL_TRANSFORM(InvDstTransform4x4)

/* QiQ_funcPtrArray (EbTransforms.h:271-296); C peer QuantizeInvQuantize (EbTransforms_C.c:89) */
SVT_AMD_API void svt_amd_QuantizeInvQuantize(int16_t *coeff, const uint32_t coeffStride, int16_t *quantCoeff,
                                             int16_t *reconCoeff, const uint32_t qFunc, const uint32_t q_offset,
                                             const int32_t shiftedQBits, const int32_t shiftedFFunc,
                                             const int32_t iq_offset, const int32_t shiftNum,
                                             const uint32_t areaSize, uint32_t *nonzerocoeff);
/* FullDistortionIntrinsic_funcPtrArray (EbPictureOperators.h:503); C peers EbPictureOperators_C.c:385-480 */
SVT_AMD_API void svt_amd_FullDistortionKernel_32bit(int16_t *coeff, uint32_t coeffStride, int16_t *reconCoeff,
                                                    uint32_t reconCoeffStride, uint64_t distortionResult[2],
                                                    uint32_t areaWidth, uint32_t areaHeight);
SVT_AMD_API void svt_amd_FullDistortionKernelCbfZero_32bit(int16_t *coeff, uint32_t coeffStride, int16_t *reconCoeff,
                                                           uint32_t reconCoeffStride, uint64_t distortionResult[2],
                                                           uint32_t areaWidth, uint32_t areaHeight);
SVT_AMD_API void svt_amd_FullDistortionKernelIntra_32bit(int16_t *coeff, uint32_t coeffStride, int16_t *reconCoeff,
                                                         uint32_t reconCoeffStride, uint64_t distortionResult[2],
                                                         uint32_t areaWidth, uint32_t areaHeight);
/* Compute8x8Satd_funcPtrArray (EbPictureOperators.h); C peers EbPictureOperators_C.c:481,563, EbHmCode.c:41,127 */
SVT_AMD_API uint64_t svt_amd_Compute8x8Satd(int16_t *diff);
SVT_AMD_API uint64_t svt_amd_Compute4x4Satd(int16_t *diff);
SVT_AMD_API uint64_t svt_amd_Compute8x8Satd_U8(uint8_t *src, uint64_t *dcValue, uint32_t srcStride);
SVT_AMD_API uint64_t svt_amd_Compute4x4Satd_U8(uint8_t *src, uint64_t *dcValue, uint32_t srcStride);
/* ResidualKernel_funcPtrArray / AdditionKernel_funcPtrArray (EbPictureOperators.h:242-330);
 * C peers ResidualKernel (:297), PictureAdditionKernel (:112) */
SVT_AMD_API void svt_amd_ResidualKernel(uint8_t *input, uint32_t inputStride, uint8_t *pred, uint32_t predStride,
                                        int16_t *residual, uint32_t residualStride, uint32_t areaWidth,
                                        uint32_t areaHeight);
SVT_AMD_API void svt_amd_PictureAdditionKernel(uint8_t *predPtr, uint32_t predStride, int16_t *residualPtr,
                                               uint32_t residualStride, uint8_t *reconPtr, uint32_t reconStride,
                                               uint32_t width, uint32_t height);

/* ------------------------------------------------------------------------- */
/* Intra prediction kernels                                                   */
/* ------------------------------------------------------------------------- */
/* mode ids of the batched form, in the order of the reference's kernel tables
 * (Codec/EbIntraPrediction.h:470-648) */
enum SvtAmdIntraKernel {
    SVT_AMD_INTRA_VERTICAL_LUMA = 0, SVT_AMD_INTRA_VERTICAL_CHROMA, SVT_AMD_INTRA_HORIZONTAL_LUMA,
    SVT_AMD_INTRA_HORIZONTAL_CHROMA, SVT_AMD_INTRA_DC_LUMA, SVT_AMD_INTRA_DC_CHROMA, SVT_AMD_INTRA_PLANAR,
    SVT_AMD_INTRA_ANGULAR_34, SVT_AMD_INTRA_ANGULAR_18, SVT_AMD_INTRA_ANGULAR_2,
    SVT_AMD_INTRA_ANGULAR_VERTICAL, SVT_AMD_INTRA_ANGULAR_HORIZONTAL
};
/* block b: reference samples at d_refs + b*ref_pitch (samples; layout [0..2N) left, [2N] top-left,
 * [2N+1..4N] top; refSampMain = block base + main_offset for the two generic angular kernels),
 * prediction written to d_pred + b*size*size (stride = size). */
SVT_AMD_API int svt_amd_intra_pred_batch(SvtAmdContext *ctx, int mode, int bytes_per_sample, int size, int skip,
                                         int32_t intraPredAngle, const void *d_refs, uint32_t ref_pitch,
                                         int32_t main_offset, void *d_pred, uint32_t nblocks);

/* LEAF forms: EB_INTRA_NOANG_TYPE / EB_INTRA_NOANG_16bit_TYPE / EB_INTRA_ANG_TYPE (EbIntraPrediction.h:430-466);
 * C peers C_DEFAULT/EbIntraPrediction_C.c:15-1051 */
#define SVT_AMD_DECL_INTRA(name, T)                                                                        \
    SVT_AMD_API void svt_amd_##name(const uint32_t size, T *refSamples, T *predictionPtr,                  \
                                    const uint32_t predictionBufferStride, const uint8_t skip);
#define SVT_AMD_DECL_INTRA_ANG(name, T)                                                                    \
    SVT_AMD_API void svt_amd_##name(uint32_t size, T *refSampMain, T *predictionPtr,                       \
                                    uint32_t predictionBufferStride, const uint8_t skip, int32_t intraPredAngle);
SVT_AMD_DECL_INTRA(IntraModeVerticalLuma, uint8_t)
SVT_AMD_DECL_INTRA(IntraModeVerticalLuma16bit, uint16_t)
SVT_AMD_DECL_INTRA(IntraModeVerticalChroma, uint8_t)
SVT_AMD_DECL_INTRA(IntraModeVerticalChroma16bit, uint16_t)
SVT_AMD_DECL_INTRA(IntraModeHorizontalLuma, uint8_t)
SVT_AMD_DECL_INTRA(IntraModeHorizontalLuma16bit, uint16_t)
SVT_AMD_DECL_INTRA(IntraModeHorizontalChroma, uint8_t)
SVT_AMD_DECL_INTRA(IntraModeHorizontalChroma16bit, uint16_t)
SVT_AMD_DECL_INTRA(IntraModeDCLuma, uint8_t)
SVT_AMD_DECL_INTRA(IntraModeDCLuma16bit, uint16_t)
SVT_AMD_DECL_INTRA(IntraModeDCChroma, uint8_t)
SVT_AMD_DECL_INTRA(IntraModeDCChroma16bit, uint16_t)
SVT_AMD_DECL_INTRA(IntraModePlanar, uint8_t)
SVT_AMD_DECL_INTRA(IntraModePlanar16bit, uint16_t)
SVT_AMD_DECL_INTRA(IntraModeAngular_34, uint8_t)
SVT_AMD_DECL_INTRA(IntraModeAngular16bit_34, uint16_t)
SVT_AMD_DECL_INTRA(IntraModeAngular_18, uint8_t)
SVT_AMD_DECL_INTRA(IntraModeAngular16bit_18, uint16_t)
SVT_AMD_DECL_INTRA(IntraModeAngular_2, uint8_t)
SVT_AMD_DECL_INTRA(IntraModeAngular16bit_2, uint16_t)
SVT_AMD_DECL_INTRA_ANG(IntraModeAngular_Vertical_Kernel, uint8_t)
SVT_AMD_DECL_INTRA_ANG(IntraModeAngular16bit_Vertical_Kernel, uint16_t)
SVT_AMD_DECL_INTRA_ANG(IntraModeAngular_Horizontal_Kernel, uint8_t)
SVT_AMD_DECL_INTRA_ANG(IntraModeAngular16bit_Horizontal_Kernel, uint16_t)

/* ------------------------------------------------------------------------- */
/* In-loop filters (deblocking, SAO) and 10-bit pack/unpack                   */
/* ------------------------------------------------------------------------- */
/* One 4-sample luma edge: `offset` = sample index of q0 of the first edge sample inside the plane;
 * tc/beta as derived by the reference's LCU drivers (EbDeblockingFilter.c:2222-3600, tables :33-66). */
typedef struct SvtAmdDlfLumaEdge { int32_t offset; int16_t tc, beta; uint8_t vertical, pad[3]; } SvtAmdDlfLumaEdge;
typedef struct SvtAmdDlfChromaEdge { int32_t offset; uint8_t cb_tc, cr_tc, vertical, pad; } SvtAmdDlfChromaEdge;
/* per-LCU SAO statistics, the four output arrays of GatherSaoStatisticsLcu* in one record */
typedef struct SvtAmdSaoStats {
    int32_t boDiff[32]; uint16_t boCount[32]; int32_t eoDiff[4][5]; uint16_t eoCount[4][5];
} SvtAmdSaoStats;

/* BATCHED (device pointers).  Edges of one launch must not overlap. */
SVT_AMD_API int svt_amd_dlf_luma_edges_batch(SvtAmdContext *ctx, void *d_plane, uint32_t stride,
                                             int bytes_per_sample, const SvtAmdDlfLumaEdge *d_edges, uint32_t nedges);
SVT_AMD_API int svt_amd_dlf_chroma_edges_batch(SvtAmdContext *ctx, void *d_cb, void *d_cr, uint32_t stride,
                                               int bytes_per_sample, const SvtAmdDlfChromaEdge *d_edges,
                                               uint32_t nedges);
/* WHOLE PICTURE boundary-strength derivation (device pointers): replaces the per-coding-unit calls SetBSArrayBasedOnPUBoundary
 * (Codec/EbDeblockingFilter.c:339) and SetBSArrayBasedOnTUBoundary (:472) with CalculateBSForPUBoundary (:109) that the
 * encode pass makes against its neighbour arrays (EbCodingLoop.c:3546-4450).  d_map: one entry per 8x8 block, raster
 * (mode: 1 INTER_MODE / 2 INTRA_MODE of the covering coding unit, dir: its interPredDirectionIndex, size_log2: log2 of the
 * coding-unit size, mv: predictionUnitArray[0].mv); d_cbf = pictureControlSetPtr->cbfMapArray (luma cbf per 4x4 block,
 * row pitch width / 4); ref_poc0 / 1 = refPOC of the two reference lists; d_lcu_edge: per LCU, 1 = tile left edge, 2 = tile
 * top edge.  Output = the layout svt_amd_dlf_picture reads ([lcu raster][256]). */
typedef struct SvtAmdCuMapEntry { uint8_t mode, dir, size_log2, pad; int16_t mv[2][2]; } SvtAmdCuMapEntry;
SVT_AMD_API int svt_amd_bs_picture(SvtAmdContext *ctx, const SvtAmdCuMapEntry *d_map, const uint8_t *d_cbf, uint32_t width,
                                   uint32_t height, int slice_type, uint64_t ref_poc0, uint64_t ref_poc1,
                                   const uint8_t *d_lcu_edge, uint8_t *d_bs_v, uint8_t *d_bs_h);
/* WHOLE PICTURE (device pointers, 4:2:0, in place): the state the per-LCU drivers LCUInternalAreaDLFCore /
 * LCUBoundaryDLFCore / LCUPictureEdgeDLFCore (+16bit; Codec/EbDeblockingFilter.c:2222, 2828, 3518, called per LCU from
 * EbCodingLoop.c:4600-4631) leave once every LCU has been through them.  d_y/d_cb/d_cr point at sample (0,0) of the
 * planes (strides in samples); d_bs_v / d_bs_h = pictureControlSetPtr->verticalEdgeBSArray / horizontalEdgeBSArray laid
 * out [lcu raster][256] (index = 4x4 block raster inside the 64x64 LCU, EbDeblockingFilter.h:19-22); d_qp = qpArray
 * (one byte per 8x8 block, row pitch qpStride); offsets = the picture control set's tcOffset / betaOffset / cbQpOffset /
 * crQpOffset.  width and height are multiples of 8.  Tile and picture boundaries carry strength 0 in the arrays. */
SVT_AMD_API int svt_amd_dlf_picture(SvtAmdContext *ctx, int bytes_per_sample, void *d_y, uint32_t strideY, void *d_cb,
                                    void *d_cr, uint32_t strideC, uint32_t width, uint32_t height,
                                    const uint8_t *d_bs_v, const uint8_t *d_bs_h, const uint8_t *d_qp, uint32_t qpStride,
                                    int32_t tcOffset, int32_t betaOffset, int32_t cbQpOffset, int32_t crQpOffset);
/* WHOLE PICTURE SAO application (device pointers, 4:2:0, OUT OF PLACE: d_dst[k] != d_src[k]): replaces
 * ApplySaoOffsetsPicture(16bit) -> ApplySaoOffsetsLcu(16bit) (Codec/EbEncDecProcess.c:522-757, :215-517; 16-bit :762-1330).
 * One record per LCU (raster order): SaoParameters_t (Codec/EbCodingUnit.h:137-146; type[0] luma, type[1] chroma:
 * 0 off, 1..4 edge offset 0 / 90 / 135 / 45 degrees, 5 band offset) with the two merge flags squeezed into bytes and the
 * tile-edge flags of lcuEdgeInfoPtr / tileInfoPtr that ApplySaoOffsetsLcu reads (:252-256) in edge_flags.
 * luma_on / chroma_on = pictureControlSetPtr->saoFlag[0] / [1]. */
typedef struct SvtAmdSaoLcuParams {
    uint8_t  merge_left, merge_up;
    uint8_t  edge_flags;        /* 1 tile left edge, 2 tile right edge, 4 tile top edge, 8 tile bottom edge */
    uint8_t  pad;
    uint32_t type[2];
    int32_t  offset[3][4];
    uint32_t band[3];
} SvtAmdSaoLcuParams;
SVT_AMD_API int svt_amd_sao_apply_picture(SvtAmdContext *ctx, int bytes_per_sample, const void *const d_src[3],
                                          void *const d_dst[3], uint32_t strideY, uint32_t strideC, uint32_t width,
                                          uint32_t height, const SvtAmdSaoLcuParams *d_lcus, int luma_on, int chroma_on);
/* SAO parameter decision of a whole picture from per-LCU statistics (device pointers): replaces the part of
 * SaoGenerationDecision / SaoGenerationDecision16bit after the statistics gathering (Codec/
 * EbSampleAdaptiveOffsetGenerationDecision.c:647-850, :936-1140): DetermineSaoLumaModeOffsets (:44), DetermineSaoChroma
 * ModeOffsets (:182), the reduced luma mode of temporal layers 0 / 1 (:347) and TestSaoCopyModes (:437).  Every LCU's own
 * best parameters are independent; the merge test needs the final parameters of the left and the upper LCU and runs as a
 * wavefront over anti-diagonals.  d_stats_y / cb / cr: [lcu raster] records as svt_amd_sao_gather_picture writes them;
 * d_enable: per LCU (NULL = all 1), 0 = the encode pass's shut-off conditions hold (EbCodingLoop.c:4675-4697), parameters
 * become zero; 1 = decide; 2 = d_params already holds this LCU's final parameters (an earlier call), a merge candidate only;
 * d_params: in: edge_flags (1 / 4 = no left / upper merge candidate), out: everything else; d_costs: [lcu][2] the luma and
 * chroma best costs the reference's call returns (also the work space between the two passes). */
typedef struct SvtAmdSaoDecisionParams {
    uint64_t lambda, chroma_lambda;                 /* contextPtr->fullLambda / fullChromaLambdaSao                      */
    uint32_t type_bits[6], merge_bits[2], offset_bits[8]; /* mdRateEstimationPtr->saoTypeIndexBits / MergeFlagBits / OffsetTrunUnaryBits */
    uint8_t  is_10bit, mm_sao, temporal_layer, pad; /* mm_sao = contextPtr->saoMode                                      */
} SvtAmdSaoDecisionParams;
SVT_AMD_API int svt_amd_sao_decide_picture(SvtAmdContext *ctx, const SvtAmdSaoDecisionParams *params, const SvtAmdSaoStats *d_stats_y,
                                           const SvtAmdSaoStats *d_stats_cb, const SvtAmdSaoStats *d_stats_cr, uint32_t lcu_cols,
                                           uint32_t lcu_rows, const uint8_t *d_enable, SvtAmdSaoLcuParams *d_params,
                                           int64_t *d_costs);
/* ONE LCU (host pointers, synchronous): the same decision with the neighbours' final parameters handed in (NULL = no
 * candidate), i.e. exactly the inputs SaoGenerationDecision has once its statistics are gathered. */
SVT_AMD_API int svt_amd_sao_decide_lcu(SvtAmdContext *ctx, const SvtAmdSaoDecisionParams *params, const SvtAmdSaoStats *stats_y,
                                       const SvtAmdSaoStats *stats_cb, const SvtAmdSaoStats *stats_cr,
                                       const SvtAmdSaoLcuParams *left, const SvtAmdSaoLcuParams *up, SvtAmdSaoLcuParams *out,
                                       int64_t costs[2]);
/* statistics of every LCU of a plane (raster LCU order), replaces the per-LCU SaoGenerationDecision ->
 * GatherSaoStatisticsLcu* calls (EbSampleAdaptiveOffsetGenerationDecision.c:647,936) */
SVT_AMD_API int svt_amd_sao_gather_picture(SvtAmdContext *ctx, int bytes_per_sample, const void *d_input,
                                           uint32_t inputStride, const void *d_recon, uint32_t reconStride,
                                           uint32_t width, uint32_t height, uint32_t lcu_size,
                                           int only_eo_90_45_135, SvtAmdSaoStats *d_stats);
/* plane-level 8+2 bit -> 16 bit pack (compressed = 4 two-bit samples per byte) and the inverse */
SVT_AMD_API int svt_amd_pack_plane(SvtAmdContext *ctx, const uint8_t *d_in8, uint32_t in8Stride, const uint8_t *d_inn,
                                   uint32_t innStride, int compressed, uint16_t *d_out16, uint32_t outStride,
                                   uint32_t width, uint32_t height);
SVT_AMD_API int svt_amd_unpack_plane(SvtAmdContext *ctx, const uint16_t *d_in16, uint32_t inStride, uint8_t *d_out8,
                                     uint32_t out8Stride, uint8_t *d_outn, uint32_t outnStride, uint32_t width,
                                     uint32_t height);

/* LEAF forms.  Luma4SampleEdgeDLFCore_Table / Chroma2SampleEdgeDLFCore_Table (EbDeblockingFilter.h:245-272) */
SVT_AMD_API void svt_amd_Luma4SampleEdgeDLFCore(uint8_t *edgeStartFilteredSamplePtr, uint32_t reconLumaPicStride,
                                                uint8_t isVerticalEdge, int32_t tc, int32_t beta);
SVT_AMD_API void svt_amd_Luma4SampleEdgeDLFCore16bit(uint16_t *edgeStartFilteredSamplePtr, uint32_t reconLumaPicStride,
                                                     uint8_t isVerticalEdge, int32_t tc, int32_t beta);
SVT_AMD_API void svt_amd_Chroma2SampleEdgeDLFCore(uint8_t *edgeStartSampleCb, uint8_t *edgeStartSampleCr,
                                                  uint32_t reconChromaPicStride, uint8_t isVerticalEdge, uint8_t cbTc,
                                                  uint8_t crTc);
SVT_AMD_API void svt_amd_Chroma2SampleEdgeDLFCore16bit(uint16_t *edgeStartSampleCb, uint16_t *edgeStartSampleCr,
                                                       uint32_t reconChromaPicStride, uint8_t isVerticalEdge,
                                                       uint8_t cbTc, uint8_t crTc);
/* SaoGatherFunctionTable* (EbSampleAdaptiveOffset.h:171-206); return EB_ERRORTYPE (0 = EB_ErrorNone) */
SVT_AMD_API int svt_amd_GatherSaoStatisticsLcuLossy_62x62(uint8_t *inputSamplePtr, uint32_t inputStride,
                                                          uint8_t *reconSamplePtr, uint32_t reconStride, uint32_t lcuWidth,
                                                          uint32_t lcuHeight, int32_t *boDiff, uint16_t *boCount,
                                                          int32_t eoDiff[4][5], uint16_t eoCount[4][5]);
SVT_AMD_API int svt_amd_GatherSaoStatisticsLcu_62x62_16bit(uint16_t *inputSamplePtr, uint32_t inputStride,
                                                           uint16_t *reconSamplePtr, uint32_t reconStride, uint32_t lcuWidth,
                                                           uint32_t lcuHeight, int32_t *boDiff, uint16_t *boCount,
                                                           int32_t eoDiff[4][5], uint16_t eoCount[4][5]);
SVT_AMD_API int svt_amd_GatherSaoStatisticsLcu_OnlyEo_90_45_135_Lossy(uint8_t *inputSamplePtr, uint32_t inputStride,
                                                                      uint8_t *reconSamplePtr, uint32_t reconStride,
                                                                      uint32_t lcuWidth, uint32_t lcuHeight,
                                                                      int32_t eoDiff[4][5], uint16_t eoCount[4][5]);
SVT_AMD_API int svt_amd_GatherSaoStatisticsLcu_62x62_OnlyEo_90_45_135_16bit(uint16_t *inputSamplePtr, uint32_t inputStride,
                                                                            uint16_t *reconSamplePtr, uint32_t reconStride,
                                                                            uint32_t lcuWidth, uint32_t lcuHeight,
                                                                            int32_t eoDiff[4][5], uint16_t eoCount[4][5]);
/* SaoFunctionTableEO_* / BO tables (EbSampleAdaptiveOffset.h:209-360) */
SVT_AMD_API int svt_amd_SAOApplyBO(uint8_t *reconSamplePtr, uint32_t reconStride, uint32_t saoBandPosition,
                                   int8_t *saoOffsetPtr, uint32_t lcuHeight, uint32_t lcuWidth);
SVT_AMD_API int svt_amd_SAOApplyEO_0(uint8_t *reconSamplePtr, uint32_t reconStride, uint8_t *temporalBufferLeft,
                                     int8_t *saoOffsetPtr, uint32_t lcuHeight, uint32_t lcuWidth);
SVT_AMD_API int svt_amd_SAOApplyEO_90(uint8_t *reconSamplePtr, uint32_t reconStride, uint8_t *temporalBufferUpper,
                                      int8_t *saoOffsetPtr, uint32_t lcuHeight, uint32_t lcuWidth);
SVT_AMD_API int svt_amd_SAOApplyEO_135(uint8_t *reconSamplePtr, uint32_t reconStride, uint8_t *temporalBufferLeft,
                                       uint8_t *temporalBufferUpper, int8_t *saoOffsetPtr, uint32_t lcuHeight,
                                       uint32_t lcuWidth);
SVT_AMD_API int svt_amd_SAOApplyEO_45(uint8_t *reconSamplePtr, uint32_t reconStride, uint8_t *temporalBufferLeft,
                                      uint8_t *temporalBufferUpper, int8_t *saoOffsetPtr, uint32_t lcuHeight,
                                      uint32_t lcuWidth);
SVT_AMD_API int svt_amd_SAOApplyBO16bit(uint16_t *reconSamplePtr, uint32_t reconStride, uint32_t saoBandPosition,
                                        int8_t *saoOffsetPtr, uint32_t lcuHeight, uint32_t lcuWidth);
SVT_AMD_API int svt_amd_SAOApplyEO_0_16bit(uint16_t *reconSamplePtr, uint32_t reconStride, uint16_t *temporalBufferLeft,
                                           int8_t *saoOffsetPtr, uint32_t lcuHeight, uint32_t lcuWidth);
SVT_AMD_API int svt_amd_SAOApplyEO_90_16bit(uint16_t *reconSamplePtr, uint32_t reconStride, uint16_t *temporalBufferUpper,
                                            int8_t *saoOffsetPtr, uint32_t lcuHeight, uint32_t lcuWidth);
SVT_AMD_API int svt_amd_SAOApplyEO_135_16bit(uint16_t *reconSamplePtr, uint32_t reconStride, uint16_t *temporalBufferLeft,
                                             uint16_t *temporalBufferUpper, int8_t *saoOffsetPtr, uint32_t lcuHeight,
                                             uint32_t lcuWidth);
SVT_AMD_API int svt_amd_SAOApplyEO_45_16bit(uint16_t *reconSamplePtr, uint32_t reconStride, uint16_t *temporalBufferLeft,
                                            uint16_t *temporalBufferUpper, int8_t *saoOffsetPtr, uint32_t lcuHeight,
                                            uint32_t lcuWidth);
/* Pack2D_funcPtrArray_16Bit_SRC ... UnPackAvg_funcPtrArray (EbPackUnPack.h:27-175) */
SVT_AMD_API void svt_amd_EB_ENC_msbPack2D(uint8_t *in8BitBuffer, uint32_t in8Stride, uint8_t *innBitBuffer,
                                          uint16_t *out16BitBuffer, uint32_t innStride, uint32_t outStride,
                                          uint32_t width, uint32_t height);
SVT_AMD_API void svt_amd_CompressedPackmsb(uint8_t *in8BitBuffer, uint32_t in8Stride, uint8_t *innBitBuffer,
                                           uint16_t *out16BitBuffer, uint32_t innStride, uint32_t outStride,
                                           uint32_t width, uint32_t height);
SVT_AMD_API void svt_amd_CPack_C(const uint8_t *innBitBuffer, uint32_t innStride, uint8_t *inCompnBitBuffer,
                                 uint32_t outStride, uint8_t *localCache, uint32_t width, uint32_t height);
SVT_AMD_API void svt_amd_EB_ENC_msbUnPack2D(uint16_t *in16BitBuffer, uint32_t inStride, uint8_t *out8BitBuffer,
                                            uint8_t *outnBitBuffer, uint32_t out8Stride, uint32_t outnStride,
                                            uint32_t width, uint32_t height);
SVT_AMD_API void svt_amd_UnPack8BitData(uint16_t *in16BitBuffer, uint32_t inStride, uint8_t *out8BitBuffer,
                                        uint32_t out8Stride, uint32_t width, uint32_t height);
SVT_AMD_API void svt_amd_UnpackAvg(uint16_t *ref16L0, uint32_t refL0Stride, uint16_t *ref16L1, uint32_t refL1Stride,
                                   uint8_t *dstPtr, uint32_t dstStride, uint32_t width, uint32_t height);

/* ------------------------------------------------------------------------- */
/* Coefficient rate estimation (mode-decision full loop / encode pass)        */
/* ------------------------------------------------------------------------- */
/* Same layout as the reference's CabacCost_t (Codec/EbCabacContextModel.h:216-226), filled on the host by
 * PrecomputeCabacCost (Codec/EbCoeffEstimation_Intrinsic.c:161) from the CABAC context state. */
typedef struct SvtAmdCabacCost {
    uint32_t CabacBitsLast[60 * 2 + 28 * 2];
    uint8_t  CabacBitsSig[2 * 42];
    uint8_t  CabacBitsG1[2 * 24];
    uint8_t  CabacBitsG2[2 * 6];
    uint8_t  CabacBitsSigMl[2 * 4];
    uint16_t CabacBitsG1x[24 / 4 * 16];
    uint8_t  CabacBitsSigV[32][16];
} SvtAmdCabacCost;
/* per-TU side information of the batched form */
typedef struct SvtAmdTuInfo {
    uint32_t num_nonzero;      /* numNonZeroCoeffs: the true count of non-zero coefficients (0 -> 0 bits) */
    uint8_t  type;             /* INTER_MODE 1 / INTRA_MODE 2 (Codec/EbDefinitions.h:345)                 */
    uint8_t  intra_luma_mode, intra_chroma_mode, component; /* component: 0 luma, else chroma            */
} SvtAmdTuInfo;
/* BATCHED: bits of nblocks contiguous size x size blocks (block b at d_coeff + b*size*size); d_bits[b] = the amount
 * EstimateQuantizedCoefficients_Lossy (EbCoeffEstimation_Intrinsic.c:1415) adds to *coeffBitsLong for that TU.
 * `cost` is a HOST pointer (copied to the device by the call). */
SVT_AMD_API int svt_amd_coeff_bits_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, uint32_t size,
                                         const int16_t *d_coeff, const SvtAmdTuInfo *d_info, uint64_t *d_bits,
                                         uint32_t nblocks);
/* The CABAC-context-UPDATING estimator: EstimateQuantizedCoefficients_generic_Update / _Update_SSE2 (Codec/EbEntropyCoding.c:2986,
 * 9355; table EstimateQuantizedCoefficientsUpdate, EbEntropyCoding.h:387), used by the mode decision's full loops when
 * coeffCabacUpdate is on (EbEncDecProcess.c:2115-2123).  Context models are SVT_AMD_COEFF_CTX_WORDS 32-bit words in
 * CoeffCtxtMdl_t's order (EbCabacContextModel.h:204-214): lastSigX[30] lastSigY[30] sig[42] coeffGroupSig[4] greater1[24]
 * greater2[6]; every call leaves the updated states in place.
 * BATCHED: nblocks contiguous size x size blocks; consecutive runs of chain_len blocks share model d_ctx_models[run] and are
 * walked in order (the units of one candidate).  d_bits[b] = the amount added to *coeffBitsLong (15 fractional bits). */
#define SVT_AMD_COEFF_CTX_WORDS 136
SVT_AMD_API int svt_amd_coeff_bits_update_batch(SvtAmdContext *ctx, uint32_t size, const int16_t *d_coeff, const SvtAmdTuInfo *d_info,
                                                uint32_t *d_ctx_models, uint64_t *d_bits, uint32_t nblocks, uint32_t chain_len);
/* LEAF, reference signature (CoeffCtxtMdl_t * as uint32_t *) */
SVT_AMD_API int svt_amd_EstimateQuantizedCoefficients_Update(uint32_t *updatedCoeffCtxModel, SvtAmdCabacCost *CabacCost, void *cabacEncodeCtxPtr,
                                                             uint32_t size, uint32_t type, uint32_t intraLumaMode, uint32_t intraChromaMode,
                                                             int16_t *coeffBufferPtr, const uint32_t coeffStride, uint32_t componentType,
                                                             uint32_t numNonZeroCoeffs, uint64_t *coeffBitsLong);
/* LEAF: slot [1][*] of EstimateQuantizedCoefficients (Codec/EbEntropyCoding.h:334-347); cabacEncodeCtxPtr unused, as in
 * the reference; returns EB_ERRORTYPE (0 = EB_ErrorNone). */
SVT_AMD_API int svt_amd_EstimateQuantizedCoefficients_Lossy(SvtAmdCabacCost *CabacCost, void *cabacEncodeCtxPtr,
                                                            uint32_t size, uint32_t type, uint32_t intraLumaMode,
                                                            uint32_t intraChromaMode, int16_t *coeffBufferPtr,
                                                            const uint32_t coeffStride, uint32_t componentType,
                                                            uint32_t numNonZeroCoeffs, uint64_t *coeffBitsLong);

/* ------------------------------------------------------------------------- */
/* Luma full loop of one mode-decision candidate                              */
/* ------------------------------------------------------------------------- */
/* Replaces ProductFullLoop (Codec/EbFullLoop.c:185-446; caller PerformFullLoop, EbProductCodingLoop.c:4351-4460) for
 * the configuration the presets >= encMode 5 use: no RDOQ / PM-core, coefficient-domain distortion, no CABAC-context
 * update (Codec/EbEncDecProcess.c:2028-2205).  Per transform unit: EstimateTransform (EbTransforms.c:3268) ->
 * ProductUnifiedQuantizeInvQuantizeMd (EbFullLoop.c:77) -> PictureFullDistortionLuma (EbPictureOperators.c:397) ->
 * TuEstimateCoeffBitsLuma (EbEntropyCoding.c:7899) -> TuCalcCostLuma (EbRateDistortionCost.c:289).  A 64x64 CU is four
 * 32x32 transform units (TU index 1..4), smaller CUs one (TU index 0). */
typedef struct SvtAmdFullLoopIn {
    uint32_t size;              /* CU size 8 / 16 / 32 / 64                                              */
    uint32_t qp;
    uint32_t slice_type;        /* EB_PICTURE: 0 B, 1 P, 2 I, 3 IDR (selects the quantiser dead zone)    */
    uint16_t pf_mode;           /* contextPtr->pfMdMode: 0 off, 1 N2, 2 N4                               */
    uint16_t pm_core;           /* contextPtr->rdoqPmCoreMethod: 0 EB_NO_RDOQ, 2 EB_PMCORE (encMode 1..4: every 4x4 block
                                 * of levels re-decided among 100 / 70 / 50 % scalings, EbTransforms.c:2807-2950)      */
    uint32_t cand_type;         /* INTER_MODE 1 / INTRA_MODE 2                                           */
    uint32_t intra_luma_mode;
    uint32_t full_lambda;       /* contextPtr->fullLambda                                                */
    uint32_t cbf_bits[4];       /* mdRateEstimationPtr->lumaCbfBits[0], [1], [5], [6]                    */
    uint32_t ycbf;              /* candidatePtr->yCbf before the call                                    */
    uint64_t coeff_bits;        /* *yCoeffBits before the call (accumulated)                             */
    uint64_t dist[2];           /* yFullDistortion before the call (accumulated for 64x64)               */
} SvtAmdFullLoopIn;
typedef struct SvtAmdFullLoopOut {
    uint32_t nz[5];             /* yCountNonZeroCoeffs[0..4] entries this call wrote (others 0)          */
    uint32_t ycbf;              /* candidatePtr->yCbf after the call                                     */
    uint64_t coeff_bits;        /* *yCoeffBits after the call                                            */
    uint64_t dist[2];           /* yFullDistortion[DIST_CALC_RESIDUAL / PREDICTION] after the call       */
    int16_t  ydc[4];            /* candidateBuffer->yDc                                                  */
    uint16_t cand_nz[4];        /* candidateBuffer->yCountNonZeroCoeffs                                  */
} SvtAmdFullLoopOut;
/* BATCHED: candidate c has its size x size luma residual at d_residual + c*4096 (row pitch = size); the quantised
 * and the reconstructed coefficients are written in the same layout.  `cost` is a HOST pointer. */
SVT_AMD_API int svt_amd_full_loop_luma_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost,
                                             const SvtAmdFullLoopIn *d_in, const int16_t *d_residual,
                                             int16_t *d_quant, int16_t *d_recon, SvtAmdFullLoopOut *d_out,
                                             uint32_t ncand);
/* Same arguments; serves the candidates whose pm_core is EB_PMCORE (2) and leaves the others alone, as the call above leaves
 * those alone: ProductUnifiedQuantizeInvQuantizeMd -> DecoupledQuantizeInvQuantizeLoops (Codec/EbFullLoop.c:121-148,
 * EbTransforms.c:2605-2973) - a picture-level switch of encMode 1..4 (EbEncDecProcess.c:2201). */
SVT_AMD_API int svt_amd_full_loop_luma_pmcore_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost,
                                                    const SvtAmdFullLoopIn *d_in, const int16_t *d_residual,
                                                    int16_t *d_quant, int16_t *d_recon, SvtAmdFullLoopOut *d_out,
                                                    uint32_t ncand);
/* Per-call form on HOST pointers (row pitch `pitch` samples, e.g. the reference's 64-sample LCU buffers); writes back
 * only the (T >> pf) area of every transform unit, like the reference.  Blocking; used by the ProductFullLoop binding. */
SVT_AMD_API int svt_amd_full_loop_luma(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdFullLoopIn *in,
                                       const int16_t *residual, int16_t *quant, int16_t *recon, uint32_t pitch,
                                       SvtAmdFullLoopOut *out);

/* ------------------------------------------------------------------------- */
/* Distortion stage of the mode decision's fast loop                           */
/* ------------------------------------------------------------------------- */
/* Replaces, for a list of candidates whose predictions are already on the device (svt_amd_intra_pu_batch /
 * svt_amd_inter_pu_batch), the measurement ProductPerformFastLoop makes after ProductMdFastPuPrediction (Codec/
 * EbProductCodingLoop.c:2036-2078): luma SAD of the predicted block against the source block (NxMSadKernel_funcPtrArray,
 * C_DEFAULT/EbComputeSAD_C.c:147) and, with contextPtr->useChromaInformationInFastLoop, Cb SAD + Cr SAD; most-probable-mode
 * candidates skip it.  What stays with the caller: the candidate that reuses its open-loop distortion (:2042-2043), the
 * ">> 2" of zero-motion 64x64 candidates in noise LCUs (:2079-2090) and the fast-cost functions themselves (scalar).
 * 8-bit 4:2:0; offsets and strides in samples; the chroma planes may be NULL when no candidate sets flag 1. */
typedef struct SvtAmdFastLoopCand {
    int32_t src_off_y, src_off_c;      /* the coding unit's block inside the source planes                         */
    int32_t pred_off_y, pred_off_c;    /* the candidate's block inside the prediction planes                       */
    uint8_t size;                      /* 8 / 16 / 32 / 64                                                         */
    uint8_t flags;                     /* 1: with chroma (useChromaInformationInFastLoop); 2: mpmFlag, skip        */
    uint8_t pad[2];
} SvtAmdFastLoopCand;
typedef struct SvtAmdFastLoopDist { uint32_t luma, chroma; } SvtAmdFastLoopDist; /* lumaFastDistortion, chromaFastDistortion */
SVT_AMD_API int svt_amd_fast_loop_distortion_batch(SvtAmdContext *ctx, const uint8_t *d_src_y, uint32_t srcStrideY,
                                                   const uint8_t *d_src_cb, const uint8_t *d_src_cr, uint32_t srcStrideC,
                                                   const uint8_t *d_pred_y, uint32_t predStrideY, const uint8_t *d_pred_cb,
                                                   const uint8_t *d_pred_cr, uint32_t predStrideC,
                                                   const SvtAmdFastLoopCand *d_cands, uint32_t ncand, SvtAmdFastLoopDist *d_out);

/* ------------------------------------------------------------------------- */
/* Encode-pass inter prediction of prediction units                            */
/* ------------------------------------------------------------------------- */
/* Replaces EncodePassInterPrediction (Codec/EbInterPrediction.c:761-926, called per prediction unit from
 * EbCodingLoop.c:3932) with its EncodeUniPredInterpolation / EncodeBiPredInterpolation (Codec/EbMcp.c:175-250, :562-760)
 * for 8-bit 4:2:0: clamp of the quarter-sample position to the padded picture (:802-812), split into integer and
 * fractional parts for luma (1/4) and chroma (1/8), the HEVC interpolation of the three planes and, for bi-prediction,
 * the average of the two 14-bit intermediates.  Reference pictures are whole padded pictures resident in HBM. */
typedef struct SvtAmdRefPicture {
    const void *d_y, *d_cb, *d_cr;     /* device pointers to the START of the padded buffers (not to sample (0,0))  */
    uint32_t strideY, strideC;         /* in samples                                                               */
    uint32_t originX, originY;         /* luma padding: sample (0,0) is at originX + originY*strideY               */
    uint32_t width, height;            /* luma                                                                     */
} SvtAmdRefPicture;
typedef struct SvtAmdInterPuJob {
    int16_t  mv[2][2];                 /* [list][x, y] in quarter samples                                          */
    uint16_t pu_x, pu_y;               /* luma position of the unit in the picture                                 */
    uint8_t  pu_w, pu_h;               /* 8 .. 64                                                                  */
    uint8_t  pred_dir;                 /* UNI_PRED_LIST_0 0, UNI_PRED_LIST_1 1, BI_PRED 2                          */
    uint8_t  pad;
    int32_t  dst_off_y, dst_off_c;     /* sample offsets of the unit inside the destination planes                 */
} SvtAmdInterPuJob;
/* BATCHED: `jobs` is a HOST array (the call derives the per-plane interpolation lists from it and uploads them);
 * reference pictures and destination planes are device memory.  ref1 may be NULL when no job uses list 1. */
SVT_AMD_API int svt_amd_inter_pu_batch(SvtAmdContext *ctx, const SvtAmdInterPuJob *jobs, uint32_t njobs,
                                       const SvtAmdRefPicture *ref0, const SvtAmdRefPicture *ref1, uint8_t *d_pred_y,
                                       uint32_t strideY, uint8_t *d_pred_cb, uint8_t *d_pred_cr, uint32_t strideC);
/* The 16-bit twin: replaces EncodePassInterPrediction16bit (Codec/EbInterPrediction.c:928-1110) with UniPredInterpolation16bit /
 * BiPredInterpolation16bit (Codec/EbMcp.c:249, :804); reference and destination planes hold 16-bit samples (10-bit content,
 * EbReferenceObject_t.referencePicture16bit), strides and offsets in samples. */
SVT_AMD_API int svt_amd_inter_pu_batch16bit(SvtAmdContext *ctx, const SvtAmdInterPuJob *jobs, uint32_t njobs,
                                            const SvtAmdRefPicture *ref0, const SvtAmdRefPicture *ref1, uint16_t *d_pred_y,
                                            uint32_t strideY, uint16_t *d_pred_cb, uint16_t *d_pred_cr, uint32_t strideC);
/* The mode decision of a 10-bit encode predicts in 8 bits from the 8 MSBs of the 16-bit reference pictures (Inter2Nx2NPuPredictionHevc
 * with is16bit: UnPackReferenceBlock, Codec/EbInterPrediction.c:414-457, 589-760): ref0 / ref1 hold 16-bit planes, the prediction is
 * 8-bit; otherwise as svt_amd_inter_pu_batch. */
SVT_AMD_API int svt_amd_inter_pu_batch_msb(SvtAmdContext *ctx, const SvtAmdInterPuJob *jobs, uint32_t njobs, const SvtAmdRefPicture *ref0,
                                           const SvtAmdRefPicture *ref1, uint8_t *d_pred_y, uint32_t strideY, uint8_t *d_pred_cb,
                                           uint8_t *d_pred_cr, uint32_t strideC);

/* ------------------------------------------------------------------------- */
/* Encode-pass intra prediction of a prediction unit from its neighbours      */
/* ------------------------------------------------------------------------- */
/* Replaces the pair GenerateIntraReferenceSamplesEncodePass (Codec/EbIntraPrediction.c:212-757; 16-bit twin :760-1300)
 * + EncodePassIntraPrediction (:4395-4673; 16-bit :4680-4960) that the encode pass reaches through the global tables
 * GenerateIntraReferenceSamplesFuncTable / EncodePassIntraPredictionFuncTable (Codec/EbCodingLoop.c:1814, :1832):
 * availability of every 4-sample neighbour group (array bound, z-order, slice, picture / tile edge, constrained intra),
 * substitution of the missing ones, [1 2 1] / strong (bilinear) smoothing of the luma reference, filtered-or-not choice by
 * mode and size (intraLumaFilterTable), and the prediction of the luma block and, with the derived chroma mode, of the
 * two chroma blocks (4:2:0).  The job carries the slices of the reference's neighbour arrays the unit can see, already
 * cut out by the caller: entry i of left[] / top[] = i-th sample below / right of the unit's top-left corner. */
typedef struct SvtAmdIntraPuJob {
    uint32_t size;                         /* 8 / 16 / 32; 4 = a luma partition of an intra 4x4 coding unit (luma only; */
                                           /* its chroma pair is the chroma part of a size-8 job at the coding unit)  */
    uint8_t  constrained_intra, strong_smoothing;
    uint8_t  pic_left, pic_top, pic_right; /* pictureLeft/Top/RightBoundary arguments (tile edges)                */
    uint8_t  bottom_left_ok, top_right_ok; /* isBottomLeftAvailable / isUpperRightAvailable(cuDepth, cuIndex)     */
    uint8_t  luma_mode, chroma_mode;       /* EB_INTRA_* / EB_INTRA_CHROMA_* (4 = derived from luma)              */
    uint8_t  mode_tl;                      /* mode-type neighbour entries: 1 INTER, 2 INTRA, 0xFF invalid,        */
    uint8_t  mode_left[16], mode_top[16];  /*   0xFE beyond the array; one per 4 luma samples                     */
    uint8_t  no_smoothing;                 /* 1: never take the filtered reference (IntraPredictionOl, the open-loop  */
    uint8_t  pad;                          /*    mode decision: neighbours = source samples, 128 where there are none) */
    uint16_t left[3][64], top[3][64];      /* [Y, Cb, Cr][i]: reconstructed neighbour samples (chroma: size used)  */
    uint16_t tl[3], pad2;
    int32_t  dst_off_y, dst_off_c;         /* sample offsets of the unit inside the three destination planes      */
} SvtAmdIntraPuJob;
/* BATCHED (device pointers): job j predicts into d_pred_y / d_pred_cb / d_pred_cr at its offsets; strides in samples. */
SVT_AMD_API int svt_amd_intra_pu_batch(SvtAmdContext *ctx, int bytes_per_sample, const SvtAmdIntraPuJob *d_jobs,
                                       uint32_t njobs, void *d_pred_y, uint32_t strideY, void *d_pred_cb, void *d_pred_cr,
                                       uint32_t strideC);
/* Per-call form on HOST pointers (one unit, blocking): the binding of the two table slots, and of the mode decision's
 * IntraPredictionCl (Codec/EbIntraPrediction.c:3682; generators GenerateIntraLuma/ChromaReferenceSamplesMd, EbProductCodingLoop.c:
 * 269, :2196), which asks for the luma block and the chroma pair separately: pred_y == NULL or pred_cb == pred_cr == NULL
 * leaves that part out.  Its open-loop twin IntraPredictionOl (:5427; neighbours = source samples cut by UpdateNeighborSamples
 * ArrayOL / UpdateChromaNeighborSamplesArrayOL, :4952, :5065: mid-grey where the picture ends, no substitution, no smoothing) is
 * the same call with every neighbour group marked available, the literal samples in left / top / tl and no_smoothing = 1. */
SVT_AMD_API int svt_amd_intra_pu(SvtAmdContext *ctx, int bytes_per_sample, const SvtAmdIntraPuJob *job, void *pred_y,
                                 uint32_t strideY, void *pred_cb, void *pred_cr, uint32_t strideC);

/* ------------------------------------------------------------------------- */
/* Device-resident encode pass: hip_encdec_segment                             */
/* ------------------------------------------------------------------------- */
/* The batched boundary of the closed-loop half (SURVEY 8b `hip_encdec_segment`) for the final encode pass: ONE call runs the
 * coding-unit loop of EncodePass (Codec/EbCodingLoop.c:2989, :3180-4594) for every LCU the host declares ready - the LCUs of
 * one wavefront step (AssignEncDecSegments, Codec/EbEncDecProcess.c:1540) - one workgroup per LCU, the units of an LCU in order
 * on the device: intra reference + prediction (GenerateIntraReferenceSamplesEncodePass, EncodePassIntraPrediction), EncodeLoop
 * (residual, Estimate transform, UnifiedQuantizeInvQuantize) and EncodeGenerateRecon of the three planes, neighbour update.
 * The picture's un-deblocked reconstruction and its mode-type map stay on the device between calls (SvtAmdEncDecPicture) and
 * stand in for the reference's ep*NeighborArray set (EbPictureControlSet.h:238-246).
 *
 * EncDec INPUT contract (what the mode decision hands over per LCU: the final coding-unit tree of LargestCodingUnit_t.
 * codedLeafArrayPtr, Codec/EbCodingUnit.h:63-88, 186-222): */
#define SVT_AMD_LCU_MAX_CUS 64
/* what the inter branch of EncodePass does with a unit (EbCodingLoop.c:3817-4400), decided by the host from the mode decision's
 * outputs before the call (mergeFlag, and for merge units skipCost <= mergeCost, :3838-3882): */
#define SVT_AMD_EP_INTER_AMVP  0   /* mergeFlag == 0: EncodeLoop of every transform unit + the luma cbf decision (PictureFullDistortionLuma,
                                    * TuEstimateCoeffBitsEncDec, EncodeTuCalcCost: coded vs. zeroed luma by distortion + lambda * rate)  */
#define SVT_AMD_EP_INTER_MERGE 1   /* merge, not skipped: EncodeLoop of every transform unit, cbf = "has coefficients" (:4164-4253)        */
#define SVT_AMD_EP_INTER_SKIP  2   /* merge decided skip: no residual, the prediction is the reconstruction (:4165-4171)                  */
typedef struct SvtAmdLcuCu {
    uint8_t x, y, size;            /* origin inside the LCU and size in luma samples (GetCodedUnitStats): 8 / 16 / 32; inter units
                                    * also 64 (four 32x32 transform units, tuItr 1..4, EbCodingLoop.c:3944-3970)                 */
    uint8_t pred_mode;             /* CodingUnit_t.predictionModeFlag: 1 INTER_MODE (2Nx2N), 2 INTRA_MODE (2Nx2N)               */
    uint8_t intra_luma_mode;       /* PredictionUnit_t.intraLumaMode, EB_INTRA_PLANAR .. EB_INTRA_MODE_34                  */
    uint8_t bottom_left_ok, top_right_ok; /* isBottomLeftAvailable / isUpperRightAvailable(depth, index), EbAvailability.c */
    uint8_t qp, chroma_qp;         /* cuPtr->qp; MapChromaQp(clip(qp + cbQpOffset + sliceCbQpOffset)) (EbCodingLoop.c:3255) */
    uint8_t leaf_index;            /* index of the unit in codedLeafArrayPtr (0..84)                                       */
    uint8_t inter_dir;             /* PredictionUnit_t.interPredDirectionIndex: UNI_PRED_LIST_0 0, UNI_PRED_LIST_1 1, BI_PRED 2 */
    uint8_t inter_kind;            /* SVT_AMD_EP_INTER_*                                                                   */
    uint32_t dz_offset;            /* dead-zone override of the luma quantiser (EbCodingLoop.c:3092-3133; 0 = none)        */
    int16_t mv[2][2];              /* PredictionUnit_t.mv[list].{x, y}, quarter samples                                    */
} SvtAmdLcuCu;
typedef struct SvtAmdLcuWork {
    uint16_t lcu_x, lcu_y;         /* luma origin of the LCU in the picture                                              */
    uint8_t num_cus;               /* coded leaves (splitFlag == 0) in Z order                                           */
    uint8_t slice_type;            /* EB_PICTURE: 0 B, 1 P, 2 I                                                           */
    uint8_t temporal_layer, constrained_intra, strong_smoothing;
    uint8_t tile_left, tile_top, tile_right; /* lcuEdgeInfoPtr->tileLeft/Top/RightEdgeFlag                              */
    uint8_t pad[4];
    uint32_t full_lambda;          /* contextPtr->fullLambda of the LCU (EncDecConfigureLcu, EbEncDecProcess.c:1448): inter units  */
    uint32_t luma_cbf_bits[4];     /* mdRateEstimationPtr->lumaCbfBits[ctx], [ctx + (NUMBER_OF_CBF_CASES >> 1)] for ctx 0, 1:
                                    * {zero cbf ctx 0, zero cbf ctx 1, non-zero ctx 0, non-zero ctx 1} (EncodeTuCalcCost)         */
    uint8_t pm_core;               /* contextPtr->mdContext->rdoqPmCoreMethod == EB_PMCORE (encMode 1..4): the quantiser of every unit is
                                    * DecoupledQuantizeInvQuantizeLoops (Codec/EbTransforms.c:3009-3052): no dead-zone override, luma levels
                                    * re-decided per 4x4 block by SSE + full_lambda * rate (needs the picture's rate tables)           */
    uint8_t pad2[11];
    SvtAmdLcuCu cu[SVT_AMD_LCU_MAX_CUS];
    uint8_t src_y[64 * 64], src_cb[32 * 32], src_cr[32 * 32]; /* source samples of the LCU (enhancedPicturePtr), pitch 64 / 32 */
} SvtAmdLcuWork;
/* EncDec OUTPUT contract (what entropy coding and the loop filters need back per LCU): the TransformUnit_t fields the encode
 * loop sets (Codec/EbTransformUnit.h:18-37), LargestCodingUnit_t.quantizedCoeff (s16, 64-pitch Y + two 32-pitch chroma planes at
 * LCU-local positions) and the un-deblocked reconstruction of the LCU. */
typedef struct SvtAmdLcuCuResult {
    uint8_t cbf[3];                /* lumaCbf, cbCbf, crCbf                       */
    uint8_t only_dc[3];            /* isOnlyDc[0..2]                              */
    uint16_t nz[3];                /* nzCoefCount[0..2]                           */
} SvtAmdLcuCuResult;
typedef struct SvtAmdLcuResult {
    SvtAmdLcuCuResult cu[SVT_AMD_LCU_MAX_CUS];           /* by position in SvtAmdLcuWork.cu */
    int16_t coeff_y[64 * 64], coeff_cb[32 * 32], coeff_cr[32 * 32];
    uint8_t rec_y[64 * 64], rec_cb[32 * 32], rec_cr[32 * 32]; /* inside the picture only */
} SvtAmdLcuResult;
/* 16-bit twins (EncodePass with is16bit: 10-bit samples in uint16_t; the source of an LCU is contextPtr->inputSample16bitBuffer as
 * EncodePassPackLcu, EbCodingLoop.c:2867, leaves it; the quantiser runs at qp + QP_BD_OFFSET, :1307) */
typedef struct SvtAmdLcuWork16 {
    uint16_t lcu_x, lcu_y;
    uint8_t num_cus, slice_type, temporal_layer, constrained_intra, strong_smoothing;
    uint8_t tile_left, tile_top, tile_right;
    uint8_t pad[4];
    uint32_t full_lambda;
    uint32_t luma_cbf_bits[4];
    uint8_t pm_core;
    uint8_t pad2[11];
    SvtAmdLcuCu cu[SVT_AMD_LCU_MAX_CUS];
    uint16_t src_y[64 * 64], src_cb[32 * 32], src_cr[32 * 32];
} SvtAmdLcuWork16;
typedef struct SvtAmdLcuResult16 {
    SvtAmdLcuCuResult cu[SVT_AMD_LCU_MAX_CUS];
    int16_t coeff_y[64 * 64], coeff_cb[32 * 32], coeff_cr[32 * 32];
    uint16_t rec_y[64 * 64], rec_cb[32 * 32], rec_cr[32 * 32];
} SvtAmdLcuResult16;
typedef struct SvtAmdEncDecPicture SvtAmdEncDecPicture;
SVT_AMD_API int svt_amd_encdec_picture_create(SvtAmdContext *ctx, uint16_t luma_width, uint16_t luma_height, int bytes_per_sample,
                                              SvtAmdEncDecPicture **out);
SVT_AMD_API int svt_amd_encdec_picture_begin(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic); /* new picture: nothing coded yet */
/* P / B pictures: the picture-level inputs of the inter units (EncodePassInterPrediction reads pictureControlSetPtr->refPicPtrArray[list],
 * Codec/EbInterPrediction.c:761; TuEstimateCoeffBitsEncDec reads pictureControlSetPtr->cabacCost, EbCodingLoop.c:4103) - reference
 * pictures of list 0 / 1 (DEVICE planes, whole padded buffers of the picture's size and sample width; either or - for an I picture that only
 * needs the rate tables, SvtAmdLcuWork.pm_core - both may be NULL) and the
 * coefficient-rate tables (HOST pointer, copied on the context's stream).  Call before the first LCU with inter units; holds until the
 * next call for this picture object. */
SVT_AMD_API int svt_amd_encdec_picture_set_inter(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdRefPicture *ref0,
                                                 const SvtAmdRefPicture *ref1, const SvtAmdCabacCost *cost);
SVT_AMD_API int svt_amd_encdec_picture_destroy(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic);
/* works / results: HOST arrays of n LCUs that do not depend on each other (left, top and top-right LCUs of each were encoded by
 * earlier calls); blocking.  Lanes of one context family may call concurrently for different LCUs of one picture. */
SVT_AMD_API int svt_amd_encode_lcus(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works, int n,
                                    SvtAmdLcuResult *results);

/* the same for a picture created with bytes_per_sample = 2 */
SVT_AMD_API int svt_amd_encode_lcus16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works, int n,
                                      SvtAmdLcuResult16 *results);

/* The whole picture in ONE call: works / results are HOST arrays of every LCU of the picture in raster order.  The wavefront of
 * AssignEncDecSegments (Codec/EbEncDecProcess.c:1540; an LCU starts when its left and top-right LCUs of the same tile are done) runs
 * on the device - workgroups draw LCUs in raster order and wait for their neighbours' completion flags - so the host makes no
 * scheduling decision and the picture costs one launch.  Implies svt_amd_encdec_picture_begin.  Blocking. */
SVT_AMD_API int svt_amd_encode_picture(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works, SvtAmdLcuResult *results);
SVT_AMD_API int svt_amd_encode_picture16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works, SvtAmdLcuResult16 *results);
/* The same on DEVICE arrays (the unit lists are not validated); asynchronous on the context's stream.  tiles: tiles of the picture
 * (they run side by side; sizes the persistent grid). */
SVT_AMD_API int svt_amd_encode_picture_device(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *d_works,
                                              SvtAmdLcuResult *d_results, int tiles);
/* An LCU waits only for what its units read across its border: intra units read the left / top-left / top / top-right LCU's reconstruction
 * and mode types, inter units nothing (they predict from the reference pictures).  LCUs without an intra unit - most LCUs of a P / B picture -
 * therefore start at once, on as many workgroups as the device holds.  The host-array calls above count such LCUs themselves; this device-array
 * form is told: free_lcus = number of LCUs without an intra unit. */
SVT_AMD_API int svt_amd_encode_picture_device_inter(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *d_works,
                                                    SvtAmdLcuResult *d_results, int tiles, int free_lcus);

/* Multi-GPU over tiles (tiles are independent inside a picture: EncDec never reads across a tile edge, Codec/EbEncDecProcess.c:2743-2760,
 * and the in-loop filters do not cross one either - loop_filter_across_tiles_enabled_flag is written 0, Codec/EbEntropyCoding.c:6346-6351).  A rank's
 * picture:   svt_amd_encode_picture_rect[16]  - works / results as in svt_amd_encode_picture (every LCU's unit list; tile-edge flags), only the LCUs
 *              inside `rect` (the rank's rectangle of whole tiles, svt_amd_tile_partition) are encoded, on the same device wavefront; the other
 *              LCUs' results come back zeroed;
 *            svt_amd_encdec_picture_deblock / _sao as on one GPU (the rank's rectangle comes out finished, the rest is not meaningful);
 *            svt_amd_encdec_picture_exchange   - the object's latest stage completed across ranks: own rectangle out, the others' in (one
 *              ncclAllGather, svt_amd_recon_exchange); _pack is its local half for hosts with their own transport (svt_amd_recon_pack);
 *            svt_amd_encdec_picture_reference  - the now complete picture padded: the next pictures' reference on every rank. */
SVT_AMD_API int svt_amd_encode_picture_rect(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works, SvtAmdLcuResult *results,
                                            const SvtAmdRect *rect);
SVT_AMD_API int svt_amd_encode_picture_rect16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works, SvtAmdLcuResult16 *results,
                                              const SvtAmdRect *rect);
/* the same restriction for the object's MODE-DECISION calls (svt_amd_md_encode_picture / _inter: mode decision + encode pass of the rank's rectangle in one call;
 * the rectangle's borders must be tile borders of the SvtAmdMdLcu records; the other LCUs' records come back zeroed); rect NULL = the whole picture */
SVT_AMD_API int svt_amd_encdec_picture_set_rect(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdRect *rect);
SVT_AMD_API int svt_amd_encdec_picture_exchange(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdRect *rects, int world, int rank);
/* Multi-GPU over PICTURES (SURVEY 8e last row): the pictures of a mini-GOP are owned by different ranks - the two non-reference B pictures of temporal layer 2 run at
 * the same time on two ranks - and a rank encodes the whole of its picture (svt_amd_encode_picture* / svt_amd_md_encode_picture*, deblocking, SAO); the only traffic is
 * the finished REFERENCE picture, sent by its owner once:
 *   svt_amd_encdec_picture_broadcast - ncclBroadcast of the object's latest stage from rank `root` (communicator of svt_amd_comm_init); on the other ranks the planes
 *     land in the object's final stage, svt_amd_encdec_picture_reference then pads the picture there as on its owner;
 *   svt_amd_encdec_picture_import    - the same hand-over between two picture objects of one process (logical ranks on one device, a host with its own transport):
 *     a copy on ctx's stream, ordered BEHIND whatever last wrote `from` (the event recorded behind its encode pass, filter or exchange).  It does not hold `from`'s owner
 *     back: the owner must not start its next picture on `from` before the importer's stream has passed the copy (svt_amd_lane_event_record / _wait, or
 *     svt_amd_synchronize on the importer's context);
 *   svt_amd_recon_broadcast          - the collective on caller-owned planes (whole allocations of bytes[p] bytes, equal on every rank). */
SVT_AMD_API int svt_amd_recon_broadcast(SvtAmdContext *ctx, void *const d_planes[3], const size_t bytes[3], int world, int rank, int root);
SVT_AMD_API int svt_amd_encdec_picture_broadcast(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, int world, int rank, int root);
SVT_AMD_API int svt_amd_encdec_picture_import(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdEncDecPicture *from);
SVT_AMD_API int svt_amd_encdec_picture_pack(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdRect *rects, int world, int r, void *d_slots,
                                            size_t slot_bytes, int to_slot);

/*
 * Entropy hand-off pre-scan.  What EncodeQuantizedCoefficients (Codec/EbEntropyCoding.c:1172; callers EncodeCoeff :4069,
 * EncodeTuCoeff :4210) does with a transform block BEFORE it codes a single bin is data parallel and needs nothing but the block: the mode-dependent
 * scan (:1347-1372), the coefficients re-ordered by sub-block and scan with a significance map per 4x4 sub-block (:1378-1430), the last
 * significant position (:1432-1461) - and the sign / greater-than-1 patterns its level loop derives from the same values (:1532-1600).  The device
 * holds every quantizedCoeff after the encode pass, so it hands the entropy coder these instead of 3 * W * H / 2 bytes of s16 planes:
 *   per transform block  SvtAmdCoeffScanTu (8 B): scan, last sub-block, last position, DC-only fast track (:1308), where its sub-blocks start;
 *   per 4x4 sub-block    SvtAmdCoeffScanGroup (8 B), sub-blocks lastScanSet .. 0 in CODING order (empty ones too: their coded_sub_block_flag is
 *                        coded): significance map in forward scan order, sign bits and |level| > 1 flags of the coded coefficients (CODING
 *                        order = set bits of the map from the highest position down; `sign` holds popcount(sigmap) bits right-aligned, the first
 *                        coded coefficient at bit popcount - 1; the first coded coefficient sits in bit 0 of `gt1`), index of its first level;
 *   per coefficient      its absolute level (u16), non-zero coefficients only, CODING order.
 * The CABAC loop that consumes them codes exactly the reference's bins: integration/svt_coeff_scan_consumer.h.
 * Blocks: slot c of component p = the unit at position c of SvtAmdLcuWork.cu (luma: the unit's size; chroma: half of it, 4x4 for an 8x8
 * unit); a 64x64 unit's four transform units are slots 1..4 (as in SvtAmdLcuResult.cu).  last_scan_set = -1: nothing to code (cbf 0).
 */
typedef struct SvtAmdCoeffScanTu {
    uint8_t scan_index;            /* SCAN_DIAG2 0 / SCAN_HOR2 1 / SCAN_VER2 2 (Codec/EbEntropyCodingUtil.h:43)                   */
    int8_t last_scan_set;          /* lastScanSet; -1 = no non-zero coefficient                                                   */
    uint8_t pos_last;              /* posLast: scan position of the last significant coefficient inside that sub-block            */
    uint8_t last_x, last_y;        /* lastSigXPos / lastSigYPos as EncodeLastSignificantXY takes them (swapped for SCAN_HOR2 / _VER2) */
    uint8_t dc_only;               /* numNonZeroCoeffs == 1 && coeff[0] != 0: the fast track of :1308-1344 (one group, one level)  */
    uint16_t first_group;          /* index of the block's first group in the LCU's group list                                    */
} SvtAmdCoeffScanTu;
typedef struct SvtAmdCoeffScanGroup {
    uint16_t sigmap;               /* bit k: the k-th coefficient of the sub-block in forward scan order is non-zero               */
    uint16_t sign;                 /* signFlags of :1547-1600: n = popcount(sigmap) sign bits, RIGHT-aligned: the first coded coefficient  */
                                   /* at bit n - 1, the last at bit 0 (what EncodeBypassBins(sign, n) writes first / last)                 */
    uint16_t gt1;                  /* bit i: the i-th coded coefficient has |level| > 1                                           */
    uint16_t first_level;          /* index of the sub-block's first level in the LCU's level list                                */
} SvtAmdCoeffScanGroup;
typedef struct SvtAmdCoeffScanLcu {
    uint32_t group_base, level_base; /* where the LCU's groups / levels start in the picture's lists                              */
    uint16_t groups, levels;         /* how many it has (<= 384 / <= 6144)                                                         */
    uint8_t pad[4];
    SvtAmdCoeffScanTu tu[3][SVT_AMD_LCU_MAX_CUS];       /* [Y, Cb, Cr][slot]                                                       */
} SvtAmdCoeffScanLcu;
/* One call per picture.  works / results: the encode pass's records of n_lcus LCUs - HOST arrays (device_arrays 0; uploaded) or DEVICE arrays
 * (device_arrays 1: what svt_amd_encode_picture_device / svt_amd_md_encode_picture left in HBM); work_stride / result_stride: sizeof the record
 * type (the 8- and 16-bit records share their heads, coefficients are s16 in both).  Outputs (HOST): lcus[n_lcus]; groups / levels: the
 * picture's lists, compacted in LCU order - capacities in elements (worst case 384 / 6144 per LCU); totals[2] = {groups, levels} written.
 * SVT_AMD_ERR_RESOURCES when a capacity is too small (totals then tell what is needed).  Blocking. */
SVT_AMD_API int svt_amd_coeff_scan_picture(SvtAmdContext *ctx, const void *works, size_t work_stride, const void *results, size_t result_stride,
                                           int device_arrays, int n_lcus, SvtAmdCoeffScanLcu *lcus, SvtAmdCoeffScanGroup *groups,
                                           uint32_t group_capacity, uint16_t *levels, uint32_t level_capacity, uint32_t totals[2]);

/* Deblocking behind the encode pass: when every LCU of the picture is encoded, a copy of the device picture (a second set of planes of
 * the picture object; the un-deblocked planes stay as they are) goes through the
 * picture-level boundary-strength and deblocking kernels (svt_amd_bs_picture + svt_amd_dlf_picture: the state the reference's
 * per-LCU drivers LCUInternalAreaDLFCore / LCUBoundaryDLFCore / LCUPictureEdgeDLFCore leave, Codec/EbCodingLoop.c:4600-4631) -
 * the finished reconstruction / reference picture when SAO is off.  works / results: the HOST records of all LCUs in raster
 * order (unit lists, QPs, tile edges; luma cbf of every unit).  out_*: optional HOST planes (tight pitch) that receive the
 * deblocked picture.  Blocking. */
typedef struct SvtAmdDeblockParams {
    int8_t tc_offset, beta_offset, cb_qp_offset, cr_qp_offset;   /* pictureControlSetPtr->tcOffset / betaOffset / cbQpOffset / crQpOffset */
    uint8_t slice_type;                                           /* EB_PICTURE: 0 B, 1 P, 2 I, 3 IDR */
    uint8_t pad[3];
    uint64_t ref_poc[2];                                          /* refPOC of the two lists (inter units; 0 for intra pictures) */
} SvtAmdDeblockParams;
SVT_AMD_API int svt_amd_encdec_picture_deblock(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works,
                                               const SvtAmdLcuResult *results, const SvtAmdDeblockParams *params, uint8_t *out_y,
                                               uint8_t *out_cb, uint8_t *out_cr);
SVT_AMD_API int svt_amd_encdec_picture_deblock16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works,
                                                 const SvtAmdLcuResult16 *results, const SvtAmdDeblockParams *params, uint16_t *out_y,
                                                 uint16_t *out_cb, uint16_t *out_cr);
/* SAO behind the deblocked device picture (after svt_amd_encdec_picture_deblock[16]): statistics of every LCU as the reference's
 * SaoGenerationDecision(16bit) gathers them inside EncodePass (Codec/EbCodingLoop.c:4640-4750, on the picture as it stands when the LCU is
 * done: deblocked, except the last 4 columns / rows of the LCU where a neighbour follows - taken from the un-deblocked picture the object
 * keeps), the parameter decision of the whole picture (svt_amd_sao_decide_picture) and ApplySaoOffsetsPicture (svt_amd_sao_apply_picture)
 * on the deblocked picture.  What is left in the picture object - and copied to out_* (HOST, tightly packed, NULL = not wanted) - is the
 * encoder's finished reconstruction: a reference picture is completed without leaving HBM.  params: the picture's rate inputs; enable:
 * HOST, one byte per LCU (NULL = all 1), 0 where the encode pass shuts SAO off (:4678-4707); lcu_params (HOST, optional): the decided
 * parameters of every LCU (what entropy coding needs).  Blocking. */
SVT_AMD_API int svt_amd_encdec_picture_sao(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works,
                                           const SvtAmdSaoDecisionParams *params, const uint8_t *enable, SvtAmdSaoLcuParams *lcu_params,
                                           uint8_t *out_y, uint8_t *out_cb, uint8_t *out_cr);
SVT_AMD_API int svt_amd_encdec_picture_sao16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works,
                                             const SvtAmdSaoDecisionParams *params, const uint8_t *enable, SvtAmdSaoLcuParams *lcu_params,
                                             uint16_t *out_y, uint16_t *out_cb, uint16_t *out_cr);
/* The parameter decision alone, on the picture object as it stands: after svt_amd_encdec_picture_deblock the encoder-order view as above;
 * without it the picture as encoded - contextPtr->allowEncDecMismatch pictures (Codec/EbEncDecProcess.c:2036-2054: temporal layers > 0 at
 * encMode >= 8, and at encMode 7 in 4K), which the reference neither deblocks nor SAO-filters on its side while it still decides (on the
 * un-deblocked reconstruction) and signals the parameters.  works: SvtAmdLcuWork[] or SvtAmdLcuWork16[] by the picture's sample width. */
SVT_AMD_API int svt_amd_encdec_picture_sao_decide(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const void *works,
                                                  const SvtAmdSaoDecisionParams *params, const uint8_t *enable, SvtAmdSaoLcuParams *lcu_params);
/* The picture object's latest stage (after SAO, else deblocked, else as encoded) as a padded reference picture in HBM - PadRefAndSetFlags
 * (Codec/EbEncDecProcess.c:1805: GeneratePadding[16Bit], edge replication) - in the form svt_amd_encdec_picture_set_inter of a LATER picture
 * object takes: reference pictures never leave the device.  origin_x / origin_y: luma padding (the reference: LCU size + 16 = 80; even);
 * planes are (width + 2 origin_x) samples wide, chroma half of everything.  out_*: optional HOST copies of the padded planes.  *ref stays
 * valid until the picture object is destroyed or the call is repeated on it. */
SVT_AMD_API int svt_amd_encdec_picture_reference(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, uint32_t origin_x, uint32_t origin_y,
                                                 SvtAmdRefPicture *ref, void *out_y, void *out_cb, void *out_cr);

/* debug: out == NULL arms per-LCU shader-clock sums in the encode-pass kernels (16 x u64 per LCU: prediction, encode, copy-out, units,
 * wait for neighbours, start, end, -, the prediction's four sub-phases, 4 unused), a later call with a HOST buffer of 16 * LCUs u64
 * fetches them (tools/encodepass_bench.py) */
SVT_AMD_API int svt_amd_debug_encdec_profile(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, unsigned long long *out);

/* LCUs the HOST encoded itself (units outside this revision: inter, intra 4x4, 64x64, delta-QP / masking configurations) are
 * handed to the device picture afterwards so that later LCUs find their neighbours: the un-deblocked last row and last column of
 * the LCU and the mode type of the 4x4 cells along them - what the reference's ep*ReconNeighborArray / epModeTypeNeighborArray
 * top and left entries hold when EncodePass returns (EncodePassUpdateReconSampleNeighborArrays, EbCodingLoop.c:1913). */
typedef struct SvtAmdLcuBorder {
    uint16_t lcu_x, lcu_y;
    uint8_t mode_bottom[16], mode_right[16];   /* 1 INTER_MODE / 2 INTRA_MODE per 4 luma samples, left to right / top to bottom */
    uint8_t bottom_y[64], right_y[64];         /* row lcu_y + h - 1 and column lcu_x + w - 1 (w, h = LCU size inside the picture) */
    uint8_t bottom_cb[32], right_cb[32], bottom_cr[32], right_cr[32];
} SvtAmdLcuBorder;
SVT_AMD_API int svt_amd_encdec_picture_put_borders(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuBorder *borders, int n);
typedef struct SvtAmdLcuBorder16 {
    uint16_t lcu_x, lcu_y;
    uint8_t mode_bottom[16], mode_right[16];
    uint16_t bottom_y[64], right_y[64], bottom_cb[32], right_cb[32], bottom_cr[32], right_cr[32];
} SvtAmdLcuBorder16;
SVT_AMD_API int svt_amd_encdec_picture_put_borders16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuBorder16 *borders, int n);

/* ------------------------------------------------------------------------- */
/* One transform unit of the final encode pass, end to end                     */
/* ------------------------------------------------------------------------- */
/* The product form of EncodeLoop + EncodeGenerateRecon (Codec/EbCodingLoop.c:651-1083, :1084-1243; 16-bit :1244-1797) for
 * one plane of one DCT unit with the default coefficient shape and no RDOQ / masking: PictureResidual -> EstimateTransform
 * -> UnifiedQuantizeInvQuantize -> EncodeInvTransform -> PictureAdditionKernel in ONE kernel, residual and coefficients in
 * registers.  Unit u: source block at d_src + src_off, prediction at d_rec + rec_off (overwritten by the reconstruction,
 * like the reference's in-place recon buffer); its quantised coefficients go to d_quant + u*size*size (row pitch = size),
 * the non-zero count to d_nz[u].  Bit depth 8 (1 byte per sample) or 10 (2 bytes). */
typedef struct SvtAmdEncodeUnit { int32_t src_off, rec_off; uint8_t qp, slice_type, pad[2]; uint32_t dz_offset; } SvtAmdEncodeUnit;
SVT_AMD_API int svt_amd_encode_tu_batch(SvtAmdContext *ctx, int bytes_per_sample, int size, const SvtAmdEncodeUnit *d_units,
                                        const void *d_src, uint32_t srcStride, void *d_rec, uint32_t recStride,
                                        int16_t *d_quant, uint32_t *d_nz, uint32_t nunits);

/* ------------------------------------------------------------------------- */
/* Quantiser of the final encode pass                                          */
/* ------------------------------------------------------------------------- */
/* Replaces UnifiedQuantizeInvQuantize (Codec/EbTransforms.c:2978-3250, called from EncodeLoop / EncodeLoop16bit,
 * Codec/EbCodingLoop.c:740-954, :1333-1550) on its paths without RDOQ / PM-core and without perceptual masking
 * (pmpMaskingLevelEncDec == 0; masking needs -brr 1): scaling constants from qp / bit depth / size / slice type
 * (:3097-3112), dead-zone override dZoffset (:3118), the DC-only shape (:3043-3090), the (size >> shape) active area,
 * the isolated-coefficient clean-up (:3190-3232) and UpdateQiQCoef (C_DEFAULT/EbTransforms_C.c:209) with the contouring
 * and forced-cbf flags.  Unit u: coefficients at d_coeff + u*1024 (row pitch = size); quantised / reconstructed
 * coefficients are written in the same layout (only the active area is touched), non-zero count to d_nz[u]. */
typedef struct SvtAmdQuantUnit {
    uint8_t  size;              /* 4 / 8 / 16 / 32                                                         */
    uint8_t  qp, bit_depth;     /* 8 or 10                                                                 */
    uint8_t  slice_type;        /* EB_PICTURE: 0 B, 1 P, 2 I, 3 IDR                                        */
    uint8_t  shape;             /* transCoeffShape: 0 default, 1 N2, 2 N4, 3 only DC                       */
    uint8_t  clean_sparse, enable_cb_flag, contouring_flag;
    uint8_t  component;         /* COMPONENT_LUMA 0, chroma otherwise                                      */
    uint8_t  temporal_layer;
    uint8_t  pad[2];
    uint32_t dz_offset;
} SvtAmdQuantUnit;
SVT_AMD_API int svt_amd_unified_quantize_batch(SvtAmdContext *ctx, const SvtAmdQuantUnit *d_units, const int16_t *d_coeff,
                                               int16_t *d_quant, int16_t *d_recon, uint32_t *d_nz, uint32_t nunits);
/* Per-call form on HOST pointers (row pitch coeffStride for all three blocks); blocking. */
SVT_AMD_API int svt_amd_unified_quantize(SvtAmdContext *ctx, const SvtAmdQuantUnit *unit, const int16_t *coeff,
                                         uint32_t coeffStride, int16_t *quant, int16_t *recon, uint32_t *nz);
/* The same reference function when contextPtr->mdContext->rdoqPmCoreMethod == EB_PMCORE (encMode 1..4; Codec/EbTransforms.c:
 * 3009-3052 -> DecoupledQuantizeInvQuantizeLoops :2605-2973): whole unit, no shapes / dead-zone override / clean-ups; luma units
 * get every 4x4 block of levels re-decided among 100 / 70 / 50 % coefficient scalings by SSE + lambda * rate (rate tables =
 * the CabacCost_t the call receives, `cost` is a HOST pointer), chroma units the plain quantiser. */
typedef struct SvtAmdPmQuantUnit {
    uint8_t  size;              /* 4 / 8 / 16 / 32                                                         */
    uint8_t  qp, bit_depth;     /* 8 or 10                                                                 */
    uint8_t  slice_type;        /* EB_PICTURE: 0 B, 1 P, 2 I, 3 IDR                                        */
    uint8_t  component;         /* COMPONENT_LUMA 0, chroma otherwise                                      */
    uint8_t  cand_type;         /* the `type` argument: INTER_MODE 1 / INTRA_MODE 2                        */
    uint8_t  pad[2];
    uint32_t lambda;            /* the `lambda` argument (contextPtr->fullLambda)                          */
} SvtAmdPmQuantUnit;
SVT_AMD_API int svt_amd_pmcore_quantize_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdPmQuantUnit *d_units,
                                              const int16_t *d_coeff, int16_t *d_quant, int16_t *d_recon, uint32_t *d_nz,
                                              uint32_t nunits);
SVT_AMD_API int svt_amd_pmcore_quantize(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdPmQuantUnit *unit,
                                        const int16_t *coeff, uint32_t coeffStride, int16_t *quant, int16_t *recon, uint32_t *nz);

/* ------------------------------------------------------------------------- */
/* Reconstruction of transform units (final encode pass)                      */
/* ------------------------------------------------------------------------- */
/* Replaces, per reconstructed plane of a transform unit, EncodeGenerateRecon / EncodeGenerateRecon16bit
 * (Codec/EbCodingLoop.c:1084-1243, :1660-1797; reached through the global table EncodeGenerateReconFunctionPtr, :1807):
 * EncodeInvTransform (Codec/EbTransforms.c:3502: DC-only shortcut, inverse DCT, inverse DST for the 4x4 luma unit) +
 * PictureAdditionKernel(16bit).  Unit u has its size x size inverse-quantised coefficients at d_coeff + u*size*size;
 * pred_off / recon_off = sample index of the unit's (0,0) in the prediction / reconstruction plane (the reference
 * reconstructs in place: both planes may be the same buffer with equal offsets). */
typedef struct SvtAmdReconUnit { int32_t pred_off, recon_off; uint8_t only_dc, dst, pad[2]; } SvtAmdReconUnit;
SVT_AMD_API int svt_amd_recon_tu_batch(SvtAmdContext *ctx, int bytes_per_sample, int size, const int16_t *d_coeff,
                                       const SvtAmdReconUnit *d_units, const void *d_pred, uint32_t predStride,
                                       void *d_recon, uint32_t reconStride, uint32_t nunits);
/* Per-call form on HOST pointers (strides in samples; coeffStride 64 / 32 for the reference's LCU scratch planes).
 * Blocking; used by the EncodeGenerateReconFunctionPtr binding. */
SVT_AMD_API int svt_amd_recon_tu(SvtAmdContext *ctx, int bytes_per_sample, int size, int only_dc, int dst,
                                 const int16_t *coeff, uint32_t coeffStride, const void *pred, uint32_t predStride,
                                 void *recon, uint32_t reconStride);

/* ------------------------------------------------------------------------- */
/* Chroma full loop of one mode-decision candidate                            */
/* ------------------------------------------------------------------------- */
/* Replaces the pair FullLoop_R (Codec/EbFullLoop.c:579-870) + CuFullDistortionFastTuMode_R (:873-1066) as the mode
 * decision calls them with PICTURE_BUFFER_DESC_CHROMA_MASK (EbProductCodingLoop.c:4291-4319 and :4518-4547), for the same
 * configuration as the luma loop above (no RDOQ / PM-core, coefficient-domain distortion, no CABAC-context update).
 * Per chroma transform unit (size/2, four 16x16 for a 64x64 CU) and plane: EstimateTransform (EbTransforms.c:3268) ->
 * UnifiedQuantizeInvQuantize_R (EbFullLoop.c:452) -> PictureFullDistortion_R (EbPictureOperators.c:325) + the chroma
 * scaling (EbFullLoop.c:1000-1004) -> TuEstimateCoeffBits_R (EbEntropyCoding.c:7963) -> TuCalcCost chroma branches
 * (EbRateDistortionCost.c:273-279).  Partial-frequency correction as :647-652 (4x4 never, 8x8 at most N2). */
typedef struct SvtAmdChromaLoopIn {
    uint32_t size;              /* CU size 8 / 16 / 32 / 64 (chroma TU = size/2, 16 for 64)              */
    uint32_t cb_qp, cr_qp;      /* MapChromaQp() outputs the caller derived                              */
    uint32_t slice_type;        /* EB_PICTURE: 0 B, 1 P, 2 I, 3 IDR                                      */
    uint32_t pf_mode;           /* contextPtr->pfMdMode: 0 off, 1 N2                                     */
    uint32_t cand_type;         /* INTER_MODE 1 / INTRA_MODE 2                                           */
    uint32_t intra_luma_mode;   /* chroma mode is EB_INTRA_CHROMA_DM in this loop                        */
    uint32_t pad;
} SvtAmdChromaLoopIn;
typedef struct SvtAmdChromaLoopOut {
    uint32_t nz[2][5];          /* countNonZeroCoeffs[1 / 2][0..4] entries the call wrote (others 0)     */
    uint32_t cbf[2];            /* candidatePtr->cbCbf / crCbf (zeroed by the caller between the two)    */
    uint64_t coeff_bits[2];     /* *cbCoeffBits / *crCoeffBits (the caller passes zeros in)              */
    uint64_t dist[2][2];        /* cb / cr FullDistortion[DIST_CALC_RESIDUAL / PREDICTION]               */
} SvtAmdChromaLoopOut;
/* BATCHED: candidate c has its (size/2)^2 Cb residual at d_residual + c*2048 and its Cr residual 1024 samples later
 * (row pitch = size/2); quantised and reconstructed coefficients come back in the same layout. `cost`: HOST pointer. */
SVT_AMD_API int svt_amd_full_loop_chroma_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost,
                                               const SvtAmdChromaLoopIn *d_in, const int16_t *d_residual,
                                               int16_t *d_quant, int16_t *d_recon, SvtAmdChromaLoopOut *d_out,
                                               uint32_t ncand);
/* Per-call form on HOST pointers ([0] = Cb, [1] = Cr; row pitch `pitch` samples, e.g. the reference's 32-sample chroma
 * LCU buffers); writes back only the (T >> pf) area of every transform unit.  Blocking. */
SVT_AMD_API int svt_amd_full_loop_chroma(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdChromaLoopIn *in,
                                         const int16_t *const residual[2], int16_t *const quant[2],
                                         int16_t *const recon[2], uint32_t pitch, SvtAmdChromaLoopOut *out);
/* coeffCabacUpdate forms (the mode decision of full-depth pictures: EbEncDecProcess.c:2115-2123; call sites EbFullLoop.c:265-280,
 * 417-432, 1014-1035): the coefficient bits of every unit come from the context-updating estimator and the candidate's
 * CoeffCtxtMdl_t (candidateBuffer->candBuffCoeffCtxModel; SVT_AMD_COEFF_CTX_WORDS words, see svt_amd_coeff_bits_update_batch) is
 * updated in place, unit after unit (chroma: Cb then Cr of every unit). */
SVT_AMD_API int svt_amd_full_loop_luma_cabac_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdFullLoopIn *d_in,
                                                   const int16_t *d_residual, int16_t *d_quant, int16_t *d_recon,
                                                   SvtAmdFullLoopOut *d_out, uint32_t *d_ctx_models, uint32_t ncand);
SVT_AMD_API int svt_amd_full_loop_chroma_cabac_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdChromaLoopIn *d_in,
                                                     const int16_t *d_residual, int16_t *d_quant, int16_t *d_recon,
                                                     SvtAmdChromaLoopOut *d_out, uint32_t *d_ctx_models, uint32_t ncand);
SVT_AMD_API int svt_amd_full_loop_luma_cabac(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdFullLoopIn *in,
                                             const int16_t *residual, int16_t *quant, int16_t *recon, uint32_t pitch,
                                             uint32_t *ctx_model, SvtAmdFullLoopOut *out);
SVT_AMD_API int svt_amd_full_loop_chroma_cabac(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdChromaLoopIn *in,
                                               const int16_t *const residual[2], int16_t *const quant[2], int16_t *const recon[2],
                                               uint32_t pitch, uint32_t *ctx_model, SvtAmdChromaLoopOut *out);

/* ------------------------------------------------------------------------- */
/* HEVC motion-compensation interpolation (closed-loop inter prediction)      */
/* ------------------------------------------------------------------------- */
/* One prediction block: integer position = sample index ref_off of the reference plane, fractional part (fx,fy) in
 * quarter samples for luma (0..3), eighth samples for chroma (0..7); output at sample index dst_off.
 * Raw (int16, bi-prediction intermediate) output is packed with stride = w at dst_off.  w, h <= 64. */
typedef struct SvtAmdMcpBlock { int32_t ref_off, dst_off; uint16_t w, h; uint8_t fx, fy, pad[2]; } SvtAmdMcpBlock;
typedef struct SvtAmdBiPredBlock { int32_t l0_off, l1_off, dst_off; uint16_t w, h; } SvtAmdBiPredBlock;
/* BATCHED (device pointers): replaces the per-PU calls of UniPredHevcInterpolationMd / EncodeUniPredInterpolation
 * (Codec/EbMcp.c:99-600) into uniPredLumaIFFunctionPtrArrayNew / uniPredChromaIFFunctionPtrArrayNew (out_raw 0) and of
 * BiPredHevcInterpolationMd (:601-1016) into biPredLumaIFFunctionPtrArrayNew / biPredChromaIFFunctionPtrArrayNew
 * (out_raw 1) followed by biPredClippingFuncPtrArray (Codec/EbMcpTables.c:14-745). */
SVT_AMD_API int svt_amd_mcp_batch(SvtAmdContext *ctx, int bytes_per_sample, int chroma, int out_raw, const void *d_ref,
                                  uint32_t refStride, void *d_dst, uint32_t dstStride,
                                  const SvtAmdMcpBlock *d_blocks, uint32_t nblocks);
/* the same with a promise about the largest block width / height of the list (0 = unknown, up to 64): smaller blocks
 * need less LDS, so more of them are in flight per CU */
SVT_AMD_API int svt_amd_mcp_batch_sized(SvtAmdContext *ctx, int bytes_per_sample, int chroma, int out_raw, const void *d_ref,
                                        uint32_t refStride, void *d_dst, uint32_t dstStride,
                                        const SvtAmdMcpBlock *d_blocks, uint32_t nblocks, uint32_t max_block_dim);
/* offset: Offset5 (luma) / ChromaOffset5 of Codec/EbDefinitions.h:1022-1030; ignored for 16-bit */
SVT_AMD_API int svt_amd_bipred_clip_batch(SvtAmdContext *ctx, int bytes_per_sample, const int16_t *d_l0,
                                          const int16_t *d_l1, void *d_dst, uint32_t dstStride, int32_t offset,
                                          const SvtAmdBiPredBlock *d_blocks, uint32_t nblocks);

/* LEAF forms: the distinct C_DEFAULT symbols of uniPredLumaIFFunctionPtrArrayNew[16], biPredLumaIFFunctionPtrArrayNew[16],
 * uniPredChromaIFFunctionPtrArrayNew[64], biPredChromaIFFunctionPtrArrayNew[64], their 16-bit twins and the two
 * clipping tables.  firstPassIFDst is the reference's scratch buffer; it is not written. */
#define SVT_AMD_DECL_MCP_UNI(name, T)                                                                          \
    SVT_AMD_API void svt_amd_##name(T *refPic, uint32_t srcStride, T *dst, uint32_t dstStride, uint32_t puWidth, \
                                    uint32_t puHeight, int16_t *firstPassIFDst);
#define SVT_AMD_DECL_MCP_RAW(name, T)                                                                          \
    SVT_AMD_API void svt_amd_##name(T *refPic, uint32_t srcStride, int16_t *dst, uint32_t puWidth,             \
                                    uint32_t puHeight, int16_t *firstPassIFDst);
#define SVT_AMD_DECL_MCP_CUNI(name, T)                                                                         \
    SVT_AMD_API void svt_amd_##name(T *refPic, uint32_t srcStride, T *dst, uint32_t dstStride, uint32_t puWidth, \
                                    uint32_t puHeight, int16_t *firstPassIFDst, uint32_t fracPosx, uint32_t fracPosy);
#define SVT_AMD_DECL_MCP_CRAW(name, T)                                                                         \
    SVT_AMD_API void svt_amd_##name(T *refPic, uint32_t srcStride, int16_t *dst, uint32_t puWidth,             \
                                    uint32_t puHeight, int16_t *firstPassIFDst, uint32_t fracPosx, uint32_t fracPosy);
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosaNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosaNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosaOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosaOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosbNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosbNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosbOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosbOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPoscNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPoscNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPoscOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPoscOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosdNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosdNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosdOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosdOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPoseNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPoseNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPoseOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPoseOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosfNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosfNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosfOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosfOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosgNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosgNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosgOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosgOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPoshNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPoshNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPoshOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPoshOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosiNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosiNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosiOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosiOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosjNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosjNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosjOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosjOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPoskNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPoskNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPoskOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPoskOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosnNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosnNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosnOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosnOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPospNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPospNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPospOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPospOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosqNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosqNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosqOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosqOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosrNew, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationFilterPosrNew16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosrOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationFilterPosrOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationCopy, uint8_t)
SVT_AMD_DECL_MCP_UNI(LumaInterpolationCopy16bit, uint16_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationCopyOutRaw, uint8_t)
SVT_AMD_DECL_MCP_RAW(LumaInterpolationCopyOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_CUNI(ChromaInterpolationCopy, uint8_t)
SVT_AMD_DECL_MCP_CUNI(ChromaInterpolationFilterOneD, uint8_t)
SVT_AMD_DECL_MCP_CUNI(ChromaInterpolationFilterTwoD, uint8_t)
SVT_AMD_DECL_MCP_CUNI(ChromaInterpolationCopy16bit, uint16_t)
SVT_AMD_DECL_MCP_CUNI(ChromaInterpolationFilterOneD16bit, uint16_t)
SVT_AMD_DECL_MCP_CUNI(ChromaInterpolationFilterTwoD16bit, uint16_t)
SVT_AMD_DECL_MCP_CRAW(ChromaInterpolationCopyOutRaw, uint8_t)
SVT_AMD_DECL_MCP_CRAW(ChromaInterpolationFilterOneDOutRaw, uint8_t)
SVT_AMD_DECL_MCP_CRAW(ChromaInterpolationFilterTwoDOutRaw, uint8_t)
SVT_AMD_DECL_MCP_CRAW(ChromaInterpolationCopyOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_CRAW(ChromaInterpolationFilterOneDOutRaw16bit, uint16_t)
SVT_AMD_DECL_MCP_CRAW(ChromaInterpolationFilterTwoDOutRaw16bit, uint16_t)
SVT_AMD_API void svt_amd_BiPredClipping(uint32_t puWidth, uint32_t puHeight, int16_t *list0Src, int16_t *list1Src,
                                        uint8_t *dst, uint32_t dstStride, int32_t offset);
SVT_AMD_API void svt_amd_BiPredClipping16bit(uint32_t puWidth, uint32_t puHeight, int16_t *list0Src, int16_t *list1Src,
                                             uint16_t *dst, uint32_t dstStride);


/* ------------------------------------------------------------------------- */
/* Device-resident mode decision: ModeDecisionLcu + EncodePass of whole pictures */
/* ------------------------------------------------------------------------- */
/* The closed loop of an LCU on the device (SURVEY 8a rows ModeDecisionLcu / ProductPerformFastLoop / PerformFullLoop / inter-depth
 * decision + the EncDec input contract): ONE call per picture replaces, for every LCU, what EncDecKernel's LCU loop does between
 * ModeDecisionConfigureLcu and the end of EncodePass (Codec/EbEncDecProcess.c:2893-3023):
 *   ModeDecisionLcu (Codec/EbProductCodingLoop.c:4691-5114): the coding-unit loop over the LCU's MdcLcuData_t leaf list - context
 *   generation, candidate injection (Codec/EbModeDecision.c:1795), the two fast loops (:1911; prediction + NxM SAD + fast cost,
 *   Codec/EbRateDistortionCost.c:440), PreModeDecision (EbModeDecision.c:300), the full loop of the surviving candidates (:4351;
 *   ProductFullLoop, Codec/EbFullLoop.c:185, + the full-cost function, EbRateDistortionCost.c:963), ProductFullModeDecision
 *   (EbModeDecision.c:1995), the candidate's reconstruction (PerformInverseTransformRecon, :1334), the inter-depth decision
 *   (EbFullLoop.c:1461) and the neighbour-array updates (:371) - followed by the LCU's EncodePass (the encode unit above) on the tree
 *   it decided, and the wavefront of AssignEncDecSegments over the picture, all in one launch.
 * The mode decision's neighbour arrays (pcs->md*NeighborArray[MD_NEIGHBOR_ARRAY_INDEX], EbPictureControlSet.h) live in HBM as
 * picture-sized maps of the SvtAmdEncDecPicture (candidate reconstruction, mode type, intra luma mode, depth, skip flag), the
 * ME / OIS results are read where the front half left them.
 *
 * THIS REVISION covers the pictures whose LCUs all take the ModeDecisionLcu path with luma-only candidates (chroma level 1,
 * CHROMA_MODE_BEST: no chroma in the mode decision, EbEncDecProcess.c:2056-2113), no CABAC-context update, intra 4x4 off, plain
 * quantiser, no delta-QP tools, 8-bit:
 *   I pictures (PICT_FULL84_DEPTH_MODE), closed-loop intra (intraMdOpenLoopFlag == 0) = the I pictures of encMode 8..10 at every
 *   resolution and of encMode 7 in 4K (BASELINE configs[0], and the I pictures of configs[1] / [2] / [3]);
 *   P / B pictures (SvtAmdMdInter below) with open-loop intra candidates, sub-sample motion, unrestricted motion vectors, the AMVP
 *   table generated in the mode decision and partial-frequency level 0 / 1 whose LCUs are all decided by ModeDecisionLcu (no
 *   branch-and-depth-pillar LCUs) = the non-reference B pictures of encMode 7..8 (half the pictures of a random-access encode).
 * svt_amd_md_picture_supported() says so; everything else stays with the reference code.
 * The controls below are DERIVED BY THE REFERENCE'S HOST CODE (SignalDerivationEncDecKernelOq, ProductResetModeDecision,
 * ModeDecisionConfigureLcu, the picture-analysis detectors) and are inputs here, like SvtAmdMeParams. */
typedef struct SvtAmdMdRates {             /* MdRateEstimationContext_t, field for field (Codec/EbMdRateEstimation.h:113-161)      */
    uint32_t splitFlagBits[6], skipFlagBits[6], mvpIndexBits[2], intraPartSizeBits[2], interPartSizeBits[8], predModeBits[2];
    uint32_t intraLumaBits[4], intraChromaBits[5], refPicBits[3], mvdBits[12], lumaCbfBits[10], chromaCbfBits[10], rootCbfBits[2];
    uint32_t transSubDivFlagBits[6], mergeFlagBits[2], mergeIndexBits[5], saoMergeFlagBits[2], saoTypeIndexBits[6];
    uint32_t saoOffsetTrunUnaryBits[8], interBiDirBits[8], interUniDirBits[2], pad[17];
} SvtAmdMdRates;
typedef struct SvtAmdMdPicture {
    uint16_t width, height;                /* luma                                                                               */
    uint8_t slice_type;                    /* EB_PICTURE: 0 B, 1 P, 2 I                                                          */
    uint8_t temporal_layer, is_reference, enc_mode;
    uint8_t depth_mode;                    /* ppcs->depthMode (PICT_FULL85 1, PICT_FULL84 2, ... EbDefinitions.h)                */
    uint8_t intra_md_open_loop;            /* ModeDecisionContext_t.intraMdOpenLoopFlag                                          */
    uint8_t intra_injection_method;        /* .intraInjectionMethod (EbEncDecProcess.c:2014-2026)                                */
    uint8_t limit_intra;                   /* .limitIntra                                                                        */
    uint8_t mpm_search, mpm_search_candidate; /* ConfigureMpm (Codec/EbModeDecisionProcess.c:560)                                */
    uint8_t pf_md_level, nfl_level_md, nmm_level_md;
    uint8_t full_loop_escape, single_fast_loop, coeff_cabac_update, spatial_sse_full_loop;
    uint8_t chroma_level;                  /* .chromaLevel                                                                       */
    uint8_t intra4x4_level;                /* .intra4x4Level (2 = off)                                                           */
    uint8_t rdoq_pmcore_method;            /* .rdoqPmCoreMethod (0 = plain quantiser)                                            */
    uint8_t skip_ois_8x8, cu8x8_mode, cu16x16_mode, limit_ois_to_dc_mode; /* PictureParentControlSet_t                            */
    uint8_t constrained_intra, strong_smoothing; /* encode pass: pcs->constrainedIntraFlag, scs->enableStrongIntraSmoothing      */
    uint8_t qp, chroma_qp;                 /* contextPtr->qp, ->chromaQp of every LCU (no delta-QP tools)                        */
    uint8_t intra8x8_restriction_inter_slice; /* .intra8x8RestrictionInterSlice (EbEncDecProcess.c:2125-2151)                    */
    uint8_t pad;
    uint32_t fast_lambda, full_lambda, fast_chroma_lambda, full_chroma_lambda; /* ModeDecisionConfigureLcu's assignment           */
    SvtAmdMdRates rates;                   /* contextPtr->mdRateEstimationPtr (slice type + QP row of mdRateEstimationArray)     */
} SvtAmdMdPicture;
#define SVT_AMD_MD_LEAVES 85               /* CU_MAX_COUNT                                                                       */
typedef struct SvtAmdMdLcu {
    uint8_t leaf_count;                    /* MdcLcuData_t.leafCount, .leafDataArray[] (EbPictureControlSet.h:82-102)            */
    uint8_t leaf_index[SVT_AMD_MD_LEAVES], leaf_split[SVT_AMD_MD_LEAVES];
    uint8_t tile_left, tile_top, tile_right; /* lcuEdgeInfoPtr                                                                   */
    uint8_t is_complete;                   /* scs->lcuParamsArray[lcu].isCompleteLcu                                             */
    uint8_t complexity_status_2;           /* ppcs->complexLcuArray[lcu] == LCU_COMPLEXITY_STATUS_2                              */
    uint8_t contouring_class[4];           /* DeriveContouringClass of the four 32x32 quadrants (EbModeDecisionConfiguration.c:395) */
    uint8_t chroma_encode_mode;            /* lcuPtr->chromaEncodeMode after ConfigureChroma (1 = CHROMA_MODE_FULL, 2 = CHROMA_MODE_BEST)             */
    uint8_t restrict_intra_global_motion;  /* contextPtr->restrictIntraGlobalMotion                                              */
    uint8_t lcu_md_mode;                   /* ppcs->lcuMdModeArray[lcu] (PICT_LCU_SWITCH pictures)                               */
    uint8_t skip_small_cu;                 /* SkipSmallCu's LCU condition (EbProductCodingLoop.c:2301): auraStatus == AURA_STATUS_0 and no
                                            * stationary edge over time                                                          */
    uint8_t cmplx_noise;                   /* ppcs->cmplxStatusLcu[lcu] == CMPLX_NOISE (:2079)                                   */
    uint8_t variance_below_200;            /* ppcs->variance[lcu][0] < 200: the encode pass's skip-cost bias (EbCodingLoop.c:3866) */
    uint8_t edge_block;                    /* ppcs->edgeResultsPtr[lcu].edgeBlockNum != 0                                        */
    uint8_t no_stop_split;                 /* StopSplitCondition's LCU exemptions (Codec/EbFullLoop.c:1445-1450): isolated non-homogeneous
                                            * area, or aura status 1 below 4K                                                    */
    uint8_t pad[3];
} SvtAmdMdLcu;
/* P / B pictures: what the inter candidates need beyond the above.  Reference pictures and coefficient-rate tables are those of
 * svt_amd_encdec_picture_set_inter (the mode decision predicts from the same reference pictures as the encode pass). */
typedef struct SvtAmdTmvpLcu {             /* TmvpUnit_t of one LCU (Codec/EbAdaptiveMotionVectorPrediction.h:28): 16 units of 16x16 */
    int16_t mv[2][16][2];                  /* [list][unit]{x, y}                                                                 */
    uint64_t ref_poc[2][16];
    uint8_t pred_dir[16], available[16];
} SvtAmdTmvpLcu;
typedef struct SvtAmdMdInter {
    uint64_t picture_number;               /* pcs->pictureNumber                                                                 */
    uint64_t ref_poc[2];                   /* ((EbReferenceObject_t *)pcs->refPicPtrArray[list]->objectPtr)->refPOC              */
    uint64_t colocated_poc;                /* refPOC of the co-located picture (list pcs->colocatedPuRefList; list 0 in P pictures) */
    uint8_t colocated_pu_ref_list, is_low_delay; /* pcs->colocatedPuRefList, ->isLowDelay                                        */
    uint8_t tmvp_enable;                   /* !ppcs->disableTmvpFlag && the co-located reference object's tmvpEnableFlag         */
    uint8_t use_subpel;                    /* ppcs->useSubpelFlag                                                                */
    uint8_t unrestricted_mv;               /* scs->staticConfig.unrestrictedMotionVector                                         */
    uint8_t generate_amvp_table_md;        /* ModeDecisionContext_t.generateAmvpTableMd                                          */
    uint8_t extra_injection;               /* .amvpInjection | .unipred3x3Injection | .bipred3x3Injection (encMode 0..1)         */
    uint8_t improve_sharpness;             /* scs->staticConfig.improveSharpness (cost biases)                                   */
    uint8_t skip_cost_bias;                /* encode pass (EbCodingLoop.c:3861-3871): non-reference B picture with a reference picture whose
                                            * intraCodedArea is above INTRA_AREA_TH[its temporal layer]                          */
    uint8_t pad[3];
    uint32_t chroma_weight;                /* ChromaWeightFactor*[qp] of the picture's class (Codec/EbRateDistortionCost.c:35-65, EbLambdaRateTables.h): the weight
                                            * of chroma distortion in the merge / skip costs                                       */
} SvtAmdMdInter;
/* what the mode decision leaves per LCU: the decision of every leaf it tested (the final tree = leaves with split == 0 walked in
 * Z order) and the costs the inter-depth decisions compared (mdLocalCuUnit[].cost) */
typedef struct SvtAmdMdLcuOut {
    uint8_t split[SVT_AMD_MD_LEAVES];      /* CodingUnit_t.splitFlag after the call                                              */
    uint8_t tested[SVT_AMD_MD_LEAVES];     /* mdLocalCuUnit[].testedCuFlag                                                       */
    uint8_t pred_mode[SVT_AMD_MD_LEAVES];  /* predictionModeFlag of tested leaves                                                */
    uint8_t intra_luma_mode[SVT_AMD_MD_LEAVES];
    uint8_t ycbf[SVT_AMD_MD_LEAVES];       /* luma cbf as ProductFullModeDecision left it: bit 0 = transformUnitArray[0].lumaCbf (units below
                                            * 64x64), bits 1..4 = transform units 1..4 of a 64x64 unit                           */
    uint8_t inter_dir[SVT_AMD_MD_LEAVES];  /* PredictionUnit_t.interPredDirectionIndex (3 in intra units)                        */
    uint8_t merge_flag[SVT_AMD_MD_LEAVES], merge_index[SVT_AMD_MD_LEAVES];
    uint8_t pad[8];
    int16_t mv[SVT_AMD_MD_LEAVES][2][2];   /* PredictionUnit_t.mv[list].{x, y}                                                   */
    uint64_t cost[SVT_AMD_MD_LEAVES];      /* mdLocalCuUnit[].cost                                                               */
    uint64_t merge_cost[SVT_AMD_MD_LEAVES], skip_cost[SVT_AMD_MD_LEAVES]; /* mdEpPipeLcu[].mergeCost / .skipCost (merge units)   */
} SvtAmdMdLcuOut;
/* 1 when this revision's device call covers the picture (see above) */
SVT_AMD_API int svt_amd_md_picture_supported(const SvtAmdMdPicture *P);
/* The whole picture: mode decision + encode pass of every LCU, one launch, wavefront on the device.  HOST arrays of every LCU in
 * raster order: lcus (controls) in; md_out (decisions), works (the EncDec input contract the decisions amount to, source samples
 * included) and results (the encode pass's output contract) out - any of the three may be NULL.  src_y / src_cb / src_cr: HOST
 * planes of the picture's source (enhancedPicturePtr), sample (0,0), strides in samples.  ois: HOST array of the picture's
 * open-loop intra search results, or NULL - then the records svt_amd_ois_picture* left in HBM for `ois_slot` are read.  cost: the
 * picture's coefficient-rate tables (pcs->cabacCost).  Blocking.
 * Source planes with rows contiguous in memory (stride_y < 2 * width + 256, multiples of 4) travel as ONE linear transfer each: the call then READS the
 * caller's inter-row padding too - stride * (rows - 1) + width bytes from src_y / src_cb / src_cr must be readable (an encoder's padded picture buffer is;
 * a plane assembled from separately allocated rows must be passed with a stride that fails the test, or copied first).
 * Round 6: the call is two kernels on the context's stream - the decisions of every LCU (wavefront on the device), then the encode pass of every LCU from the work
 * records the first left in HBM. */
SVT_AMD_API int svt_amd_md_encode_picture(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdMdPicture *P, const SvtAmdMdLcu *lcus,
                                          const uint8_t *src_y, uint32_t stride_y, const uint8_t *src_cb, const uint8_t *src_cr,
                                          uint32_t stride_c, const SvtAmdOisLcuResult *ois, int ois_slot, const SvtAmdCabacCost *cost,
                                          SvtAmdMdLcuOut *md_out, SvtAmdLcuWork *works, SvtAmdLcuResult *results);
/* P / B pictures: the same call with the inter inputs - X (above), me: HOST array of the picture's motion-estimation results (one
 * record per LCU; NULL = the records svt_amd_me_picture* left in HBM for `me_slot`), tmvp: HOST array of the co-located picture's
 * motion field (one record per LCU; may be NULL when !X->tmvp_enable).  The reference pictures and rate tables are those of the
 * last svt_amd_encdec_picture_set_inter on `pic`.  X == NULL: an I picture, as above. */
SVT_AMD_API int svt_amd_md_encode_picture_inter(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdMdPicture *P, const SvtAmdMdInter *X,
                                                const SvtAmdMdLcu *lcus, const uint8_t *src_y, uint32_t stride_y, const uint8_t *src_cb,
                                                const uint8_t *src_cr, uint32_t stride_c, const SvtAmdOisLcuResult *ois, int ois_slot,
                                                const SvtAmdMeLcuResult *me, int me_slot, const SvtAmdTmvpLcu *tmvp,
                                                SvtAmdMdLcuOut *md_out, SvtAmdLcuWork *works, SvtAmdLcuResult *results);
/* The same two calls for a 10-bit picture (a picture object created with bytes_per_sample = 2; encoderBitDepth 10, BASELINE configs[3]).  The reference's mode decision
 * stays an 8-bit process there - its source is the input picture's 8-bit plane, its inter candidates are predicted from the 8 most significant bits of the 16-bit reference
 * pictures (Inter2Nx2NPuPredictionHevc with is16bit: UnPackReferenceBlock, Codec/EbInterPrediction.c:414-457, :589-760; AddChromaEncDec likewise, EbCodingLoop.c:3841) - and
 * only EncodePass codes the 10-bit samples (EncodePassPackLcu, EbCodingLoop.c:2867).  src_*: HOST planes of the 10-bit source in 16-bit words (8-bit plane << 2 | the two
 * extra bits), strides in samples; the call derives the 8-MSB views of the source and of the picture object's 16-bit reference pictures on the device and decides on
 * those, then encodes on the 10-bit samples: works / results are the 16-bit records of svt_amd_encode_picture16.  Decisions (md_out) as in the 8-bit calls. */
SVT_AMD_API int svt_amd_md_encode_picture16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdMdPicture *P, const SvtAmdMdLcu *lcus,
                                            const uint16_t *src_y, uint32_t stride_y, const uint16_t *src_cb, const uint16_t *src_cr,
                                            uint32_t stride_c, const SvtAmdOisLcuResult *ois, int ois_slot, const SvtAmdCabacCost *cost,
                                            SvtAmdMdLcuOut *md_out, SvtAmdLcuWork16 *works, SvtAmdLcuResult16 *results);
SVT_AMD_API int svt_amd_md_encode_picture_inter16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdMdPicture *P, const SvtAmdMdInter *X,
                                                  const SvtAmdMdLcu *lcus, const uint16_t *src_y, uint32_t stride_y, const uint16_t *src_cb,
                                                  const uint16_t *src_cr, uint32_t stride_c, const SvtAmdOisLcuResult *ois, int ois_slot,
                                                  const SvtAmdMeLcuResult *me, int me_slot, const SvtAmdTmvpLcu *tmvp, SvtAmdMdLcuOut *md_out,
                                                  SvtAmdLcuWork16 *works, SvtAmdLcuResult16 *results);
SVT_AMD_API int svt_amd_md_picture_supported_inter(const SvtAmdMdPicture *P, const SvtAmdMdInter *X);
/* 1 when every one of the n LCUs is decided by ModeDecisionLcu with luma-only candidates (the per-LCU half of the two checks above) */
SVT_AMD_API int svt_amd_md_lcus_supported(const SvtAmdMdPicture *P, const SvtAmdMdLcu *lcus, int n);
/* debug: stage clocks of the mode-decision kernel (16 shader-clock sums per LCU, accumulated over the picture object's later calls; the first call
 * switches the collection on) - see svt-hevc_amd/csrc/md_kernels.hip */
SVT_AMD_API int svt_amd_debug_md_profile(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, unsigned long long *out);
SVT_AMD_API int svt_amd_debug_md_profile_sub(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, unsigned long long *out);
/* ... and both sets of sums by the depth of the coding unit they were spent on: out[LCU][depth 0..3][32] */
SVT_AMD_API int svt_amd_debug_md_profile_depth(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, unsigned long long *out);
/* debug / test: the full loops' 16x16 and 32x32 forward transforms on the register butterflies (on != 0) instead of the matrix cores for the picture object's later calls:
 * both paths must give the reference's decisions (tests/test_gpu_md.py) */
SVT_AMD_API int svt_amd_debug_md_force_butterflies(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, int on);
/* development builds (-DMD_TRACE) only, SVT_AMD_ERR_BAD_PARAM otherwise: lane-0 time stamps of the kernel's four waves along two units of one LCU (tools/md_trace.py) */
SVT_AMD_API int svt_amd_debug_md_trace(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, int lcu, int unit, unsigned long long *out);
/* measurement: duration in ms (HIP events on the call's stream) and launch width (workgroups) of the picture object's last mode-decision kernel launch */
SVT_AMD_API int svt_amd_debug_md_kernel_ms(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, float *ms, int *workgroups);
/* ... and of the encode-pass kernel queued behind it by the same call */
SVT_AMD_API int svt_amd_debug_md_ep_ms(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, float *ms);
/* diagnosis: the mode-decision launches that hold (or wait for) workgroups of the device budget right now (md_kernels.hip: MdFlight) */
SVT_AMD_API int svt_amd_debug_md_flights(int *in_flight, int *workgroups_held, int *waiting);
/* Everything the first svt_amd_md_encode_picture[_inter / 16] call on a picture object would allocate (device state, page-locked staging, events), made NOW: for hosts that
 * build their picture objects before their clock starts (the encoder binding at EbInitEncoder time) - allocations made while other pictures' kernels run wait for them. */
SVT_AMD_API int svt_amd_md_picture_warmup(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic);
/* measurement: dynamic LDS bytes a workgroup of k_md_encode_picture<inter, sample bytes> is launched with (the whole LCU state: one workgroup per CU) */
SVT_AMD_API int svt_amd_debug_md_kernel_lds_bytes(int inter, int bytes_per_sample);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HEVC_AMD_H */
