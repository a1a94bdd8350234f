/*
 * integration/svt_hook_internal.h - shared between the binding files (svt_hook_me.c: front half and per-call EncDec bindings;
 * svt_hook_encdec.c: the device-resident encode pass).  Part of the code a maintainer adds to the reference; no reference source.
 */
#ifndef SVT_HOOK_INTERNAL_H
#define SVT_HOOK_INTERNAL_H
#include <stdio.h>
#include "EbDefinitions.h"
#include "EbPictureBufferDesc.h"
#include "EbEncDecProcess.h"
#include "../include/svt_hevc_amd.h"

/* The bindings' CONFIGURATION: what the environment says where it says something, the product's defaults otherwise.  The defaults are the configuration bench.py
 * measures and the end-to-end suite runs (the benchmarked configuration is the one a host gets by loading the library): the closed loop of P / B
 * pictures on the device (SVT_HOOK_MD=pb), 16 picture control sets in the EncDec pool (SVT_HOOK_PCS_POOL), 12 EncDec lanes and 4 front-half lanes (= streams = hardware
 * queues: SVT_HOOK_EP_LANES, SVT_HOOK_FRONT_LANES).  "0" or "off" in the environment switches a binding off; any other switch is NULL unless the environment sets it. */
const char *svt_hook_cfg(const char *name);
/* the root device context (created with the encoder, or here on first use) and the loud exit every binding shares */
SvtAmdContext *svt_hook_device(uint16_t lumaWidth, uint16_t lumaHeight);
void svt_hook_die(const char *what);
/* Every mutex a binding takes goes through these two, so that the loud exit knows what the failing thread holds: svt_hook_die unlocks them (and gives the thread's
 * device lane back, svt_hook_encdec_thread_exit) BEFORE it reports and leaves the thread - the other kernel threads must not block for ever on a lock or a lane whose
 * owner is gone, or EbDeinitEncoder's pthread_join never returns.  After a failure svt_hook_failed() is non-zero and every binding falls through to the reference code. */
#include <pthread.h>
void svt_hook_lock(pthread_mutex_t *m);
void svt_hook_unlock(pthread_mutex_t *m);
int svt_hook_failed(void);
void svt_hook_encdec_thread_exit(void);
/* SVT_HOOK_TIMELINE=<file>: a time line of the bindings' picture-level events (device calls, reference uploads, first / last LCU of a picture through EncodePass), one
 * line each, written with the report - what keeps how many pictures in flight.  kind: a short tag; a, b: event-specific numbers. */
void svt_hook_timeline(const char *kind, unsigned long long picture, int a, int b, double t_begin, double t_end);
double svt_hook_now(void);
int svt_hook_timeline_enabled(void);
/* the running pipeline's application callback (error reporting): noted by the first bound call that sees the sequence control set */
struct SequenceControlSet_s;
void svt_hook_note_callback(const struct SequenceControlSet_s *scs);

/* Non-zero while the reference's EncodePass runs on this thread for an LCU the device has already encoded (svt_hook_encdec.c):
 * the leaves it reaches answer from the device's result instead of computing. */
extern __thread int svt_hook_ep_active;
/* UnifiedQuantizeInvQuantize of the served LCU: quantised coefficients, count and the DC marker of the unit / plane the call is for */
void svt_hook_ep_quantize(EncDecContext_t *contextPtr, EB_S16 *quantCoeff, EB_S16 *reconCoeff, EB_U32 coeffStride, EB_U32 qp, EB_U32 areaSize,
                          EB_U32 *nz, EB_U32 shape, EB_U32 cleanSparse, EB_U32 masking, EB_U32 enableCbflag, EB_U32 contouring, EB_U32 dZoffset);
/* EncodeGenerateRecon of the served LCU: the unit's reconstruction into the picture buffer */
void svt_hook_ep_recon(EncDecContext_t *contextPtr, EB_U32 originX, EB_U32 originY, EB_U32 tuSize, EbPictureBufferDesc_t *recon);
void svt_hook_encdec_report(FILE *out);
/* EbInitEncoder time, device up: lanes, picture objects, mode-decision state and records of every EncDec picture of the pool (svt_hook_encdec.c) */
void svt_hook_encdec_warmup(void);
/* SVT_HOOK_ENCODEPASS_REFS: a picture whose LCUs were all encoded on the device is finished there too (deblocking, SAO, padding) and becomes
 * the reference picture of later pictures without an upload.  The SAO wraps tell the encode-pass binding which LCUs the reference ran the
 * decision for and with which rate inputs; the reference cache takes the finished picture. */
#include "EbMdRateEstimation.h"
void svt_hook_ep_note_sao(const PictureControlSet_t *pcs, EB_U32 x, EB_U32 y, const MdRateEstimationContext_t *md, EB_U64 lambda, EB_U64 chromaLambda,
                          int mmSao, int is16);
/* page-locks the planes of a pooled picture buffer of the encoder once (svt_hook_me.c) */
void svt_hook_pin_picture(const EbPictureBufferDesc_t *p, size_t bps);
/* the same at pool-construction time (EbInitEncoder): now if the device context exists, at device start-up otherwise */
void svt_hook_pin_picture_at_init(const EbPictureBufferDesc_t *p, size_t bps);
void svt_hook_register_device_reference(const EbPictureBufferDesc_t *p, uint64_t poc, size_t bps, const SvtAmdRefPicture *dev);
#endif
