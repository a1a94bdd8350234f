/*
 * Internal declarations of the HIP library (not part of the C-ABI).
 * Device memory layout of one picture slot (all planes 8-bit, row pitch a
 * multiple of 256 B, sample (0,0) 128-B aligned so LCU rows are fetched as
 * aligned 64-B segments):
 *
 *   full      (W   x H  ) valid x in [-68 , W+68),  y in [-68, H+68)   PA "inputPaddedPicture"
 *   quarter   (W/2 x H/2) valid pad 32                                  "quarterDecimatedPicture"
 *   sixteenth (W/4 x H/4) valid pad 16                                  "sixteenthDecimatedPicture"
 *   hp_b / hp_h / hp_j    geometry of `full`; AVC-style half-pel planes
 */
#ifndef SVT_AMD_INTERNAL_H
#define SVT_AMD_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svt_hevc_amd.h"

struct DevPlane {
    uint8_t *origin;   /* device pointer to sample (0,0) */
    int32_t  pitch;    /* bytes per row */
    int32_t  width, height, pad;
    uint8_t *alloc;    /* base of the allocation */
    size_t   alloc_bytes;
    int32_t  lead_rows; /* rows above y = -pad kept as guard */
    int32_t  lead_cols; /* bytes left of x = 0 in each row   */
};

struct DevPicture {
    DevPlane full, quarter, sixteenth, hp_b, hp_h, hp_j;
    SvtAmdMeLcuResult *d_me_out;   /* device buffer, one record per LCU */
    SvtAmdOisLcuResult *d_ois_out; /* device buffer, one record per LCU */
    void *d_me_carry;              /* MeCarry per LCU (me_kernels.hip), 192 B reserved each */
    uint8_t *d_staging;            /* device copy of the raw luma (upload path) */
    size_t   staging_bytes;
    uint8_t *h_staging;            /* pinned host copy of the raw luma (asynchronous upload path), allocated on first use */
    uint8_t *d_pack;               /* compact wire form of this slot's ME + OIS records (svt_amd_*_fetch_compact_async) */
    size_t pack_bytes;
    hipEvent_t ev_ready;           /* recorded after the planes of this slot were built: lanes on other streams wait on it */
    hipEvent_t ev_me, ev_ois;      /* recorded behind the kernels that wrote d_me_out / d_ois_out: a consumer on another lane's stream (the mode decision reading the
                                    * records where they are) waits on them */
    uint32_t me_lcus, ois_lcus;    /* LCUs of the picture whose records the buffers hold (0: none yet, or only a part of the picture: every upload into the slot resets
                                    * them; read / written across host threads with acquire / release) */
    /* which LCUs of the slot's CURRENT picture the range launches since the last upload have covered - a bit per LCU (8K: 8,160), so bands may arrive in any order and
     * from several lanes (me_cov_lock); me_lcus is set when every LCU is covered */
    uint64_t me_cov[128];
    uint32_t me_cov_count;
    int me_cov_lock;
    hipEvent_t ev_md_read;         /* recorded behind a mode-decision kernel that reads d_me_out / d_ois_out in place: the next ME / OIS launch INTO the slot waits for it */
    int md_read_pending;
    uint16_t width, height;
    int      valid;
};

/* kernel-side view */
struct PicView {
    const uint8_t *full, *quarter, *sixteenth, *hp_b, *hp_h, *hp_j;
    int32_t pitch_full, pitch_quarter, pitch_sixteenth;
};

enum { KC_PREP = 0, KC_ME_SEARCH = 1, KC_OIS = 2, KC_COUNT = 3 };

#define SVT_AMD_MAX_BATCH 256

/* one motion-estimation job = one picture (or an LCU range of it) against its references */
struct MeJobDev {
    SvtAmdMeParams P;
    PicView cur, ref0, ref1;
    SvtAmdMeLcuResult *out;
    struct MeCarry *carry;         /* per-LCU hand-over between the HME and the search kernel */
    int32_t lcu_begin, lcu_count;
    unsigned long long *dbg_clock; /* optional: 16 clock stamps per workgroup (phase profile) */
};

/* one open-loop intra search job = one picture */
struct OisJobDev {
    SvtAmdOisParams P;
    const uint8_t *full;           /* padded source luma, sample (0,0) */
    int32_t pitch, lcus_w, nlcu;
    const SvtAmdMeLcuResult *me;   /* ME results of the picture (P/B) */
    SvtAmdOisLcuResult *out;
};

struct SvtAmdContext {
    int device;
    SvtAmdContext *parent;         /* lane (svt_amd_context_fork): shares the parent's picture slots, owns everything else */
    hipStream_t stream;
    uint16_t max_w, max_h;
    int num_slots;
    DevPicture *slots;
    hipEvent_t ev_begin, ev_end;
    /* per-kernel-class event pairs recorded while the timer is armed */
    int timer_armed;
    struct Stamp { hipEvent_t a, b; int cls; } *stamps;
    int num_stamps, cap_stamps;
    SvtAmdMeLcuResult *d_me_scratch; /* host-supplied ME results for svt_amd_ois_picture */
    void *d_prep_jobs;             /* device array of SVT_AMD_MAX_BATCH prep descriptors (128 B each reserved) */
    OisJobDev *d_ois_jobs;         /* device array of SVT_AMD_MAX_BATCH OIS job descriptors */
    MeJobDev *d_jobs;              /* device array of SVT_AMD_MAX_BATCH job descriptors */
    unsigned long long *d_dbg;     /* phase-profile buffer (svt_amd_debug_me_phase_profile) */
    size_t dbg_slots;
    void *d_cabac_cost;            /* this context's copy of the caller's CabacCost_t (rate_device.h) */
    uint8_t *d_leaf_scratch;       /* staging of the one-unit host-pointer forms (svt_amd_ctx_scratch) */
    size_t leaf_scratch_bytes;
    /* front-end pipeline (svt_amd_frontend_submit / _wait): pinned result buffers + completion event of this lane */
    SvtAmdMeLcuResult *h_me;
    SvtAmdOisLcuResult *h_ois;
    hipEvent_t ev_done;
    int frontend_busy;
    hipEvent_t ev_user[8];
    uint8_t *h_desc_ring;          /* pinned ring of launch descriptors (svt_amd_upload_descriptors) */
    hipEvent_t ev_desc[8];
    int desc_next;         /* svt_amd_lane_event_record / _wait */
    /* multi-GPU exchange (comm.hip): RCCL communicator + the all-gather buffer (one slot per rank) */
    void *comm;
    int comm_world, comm_rank;
    uint8_t *d_xchg;
    size_t xchg_bytes;
};

/* device scratch of at least `bytes` owned by the context (grown on demand, freed by svt_amd_context_destroy);
 * callers serialise per context, as for every other call on one context */
/* launch descriptors (job arrays) reach the device without the copy engines: see context.hip */
int svt_amd_upload_descriptors(SvtAmdContext *ctx, void *d_dst, const void *src, size_t bytes);
int svt_amd_ctx_scratch(SvtAmdContext *ctx, size_t bytes, uint8_t **out);
/* context.hip: the device's view of a range inside a svt_amd_host_register'ed buffer, or nullptr */
const void *svt_amd_registered_device_ptr(const void *h_ptr, size_t bytes);

void svt_amd_set_error(const char *fmt, ...);
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            svt_amd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),       \
                              __FILE__, __LINE__);                                         \
            return SVT_AMD_ERR_DEVICE;                                                     \
        }                                                                                  \
    } while (0)

/* stamps a kernel class duration when the timer is armed */
int svt_amd_stamp_begin(SvtAmdContext *ctx, int cls);
int svt_amd_stamp_end(SvtAmdContext *ctx);

/* kernel launchers (prep_kernels.hip / me_kernels.hip) */
int svt_amd_launch_prep(SvtAmdContext *ctx, DevPicture *pic, const uint8_t *d_luma, uint32_t stride);
int svt_amd_launch_prep_batch(SvtAmdContext *ctx, DevPicture *const *pics, const uint8_t *const *d_luma, uint32_t stride,
                              int n);
int svt_amd_launch_me_batch(SvtAmdContext *ctx, const MeJobDev *host_jobs, int njobs, int max_lcus);
int svt_amd_launch_zz_sad(SvtAmdContext *ctx, const DevPicture *cur, const DevPicture *prev, SvtAmdZzLcu *d_out);
int svt_amd_launch_ois_batch(SvtAmdContext *ctx, const struct OisJobDev *host_jobs, int njobs, int max_lcus);

static inline PicView make_view(const DevPicture *p)
{
    PicView v;
    v.full = p->full.origin;
    v.quarter = p->quarter.origin;
    v.sixteenth = p->sixteenth.origin;
    v.hp_b = p->hp_b.origin;
    v.hp_h = p->hp_h.origin;
    v.hp_j = p->hp_j.origin;
    v.pitch_full = p->full.pitch;
    v.pitch_quarter = p->quarter.pitch;
    v.pitch_sixteenth = p->sixteenth.pitch;
    return v;
}

#endif
