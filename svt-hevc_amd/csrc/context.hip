/*
 * Context, picture slots and the batched C-ABI entry points (include/svt_hevc_amd.h).
 * Host side only; kernels live in prep_kernels.hip / me_kernels.hip / leaf.hip.
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "svt_amd_internal.h"
#include <vector>
#include <mutex>

static thread_local char g_err[512] = "";

void svt_amd_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *svt_amd_last_error(void) { return g_err; }
extern "C" const char *svt_amd_version(void) { return "svt-hevc_amd 0.1 (gfx950)"; }

static int align_up(int v, int a) { return (v + a - 1) / a * a; }

static int plane_create(DevPlane *p, int w, int h, int pad)
{
    memset(p, 0, sizeof(*p));
    p->width = w;
    p->height = h;
    p->pad = pad;
    p->lead_cols = align_up(pad, 128);
    p->pitch = align_up(w + p->lead_cols + pad + 64, 256);
    p->lead_rows = pad + 8;
    const size_t rows = (size_t)h + 2 * (size_t)p->lead_rows;
    p->alloc_bytes = rows * (size_t)p->pitch;
    HIP_TRY(hipMalloc((void **)&p->alloc, p->alloc_bytes));
    HIP_TRY(hipMemset(p->alloc, 0, p->alloc_bytes));
    p->origin = p->alloc + (size_t)p->lead_rows * p->pitch + p->lead_cols;
    return SVT_AMD_OK;
}

static void plane_destroy(DevPlane *p)
{
    if (p->alloc)
        (void)hipFree(p->alloc);
    memset(p, 0, sizeof(*p));
}

extern "C" void svt_amd_context_destroy(SvtAmdContext *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    (void)svt_amd_comm_destroy(ctx);
    for (int i = 0; !ctx->parent && ctx->slots && i < ctx->num_slots; i++) {
        DevPicture *s = &ctx->slots[i];
        plane_destroy(&s->full);
        plane_destroy(&s->quarter);
        plane_destroy(&s->sixteenth);
        plane_destroy(&s->hp_b);
        plane_destroy(&s->hp_h);
        plane_destroy(&s->hp_j);
        if (s->d_me_out)
            (void)hipFree(s->d_me_out);
        if (s->d_ois_out)
            (void)hipFree(s->d_ois_out);
        if (s->d_me_carry)
            (void)hipFree(s->d_me_carry);
        if (s->d_staging)
            (void)hipFree(s->d_staging);
        if (s->d_pack)
            (void)hipFree(s->d_pack);
        if (s->h_staging)
            (void)hipHostFree(s->h_staging);
        if (s->ev_ready)
            (void)hipEventDestroy(s->ev_ready);
        if (s->ev_me)
            (void)hipEventDestroy(s->ev_me);
        if (s->ev_ois)
            (void)hipEventDestroy(s->ev_ois);
        if (s->ev_md_read)
            (void)hipEventDestroy(s->ev_md_read);
    }
    if (!ctx->parent)
        free(ctx->slots);
    if (ctx->d_cabac_cost)
        (void)hipFree(ctx->d_cabac_cost);
    if (ctx->d_leaf_scratch)
        (void)hipFree(ctx->d_leaf_scratch);
    if (ctx->h_me)
        (void)hipHostFree(ctx->h_me);
    if (ctx->h_ois)
        (void)hipHostFree(ctx->h_ois);
    if (ctx->h_desc_ring) {
        (void)hipHostFree(ctx->h_desc_ring);
        for (int i = 0; i < 8; i++)
            (void)hipEventDestroy(ctx->ev_desc[i]);
    }
    for (int i = 0; i < 8; i++)
        if (ctx->ev_user[i])
            (void)hipEventDestroy(ctx->ev_user[i]);
    if (ctx->ev_done)
        (void)hipEventDestroy(ctx->ev_done);
    if (ctx->d_jobs)
        (void)hipFree(ctx->d_jobs);
    if (ctx->d_me_scratch)
        (void)hipFree(ctx->d_me_scratch);
    if (ctx->d_ois_jobs)
        (void)hipFree(ctx->d_ois_jobs);
    if (ctx->d_prep_jobs)
        (void)hipFree(ctx->d_prep_jobs);
    if (ctx->d_dbg)
        (void)hipFree(ctx->d_dbg);
    for (int i = 0; i < ctx->cap_stamps; i++) {
        (void)hipEventDestroy(ctx->stamps[i].a);
        (void)hipEventDestroy(ctx->stamps[i].b);
    }
    free(ctx->stamps);
    if (ctx->ev_begin)
        (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end)
        (void)hipEventDestroy(ctx->ev_end);
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    free(ctx);
}

/* stream, events, descriptor / cost buffers: everything a context or a lane owns besides the picture slots */
static int context_common_create(SvtAmdContext *ctx)
{
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&ctx->ev_begin));
    HIP_TRY(hipEventCreate(&ctx->ev_end));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming | hipEventBlockingSync)); /* waited on for milliseconds by several host threads: they sleep */
    const int nlcu = ((ctx->max_w + 63) / 64) * ((ctx->max_h + 63) / 64);
    if (hipMalloc((void **)&ctx->d_me_scratch, (size_t)nlcu * sizeof(SvtAmdMeLcuResult)) != hipSuccess ||
        hipMalloc(&ctx->d_prep_jobs, 128 * SVT_AMD_MAX_BATCH) != hipSuccess ||
        hipMalloc((void **)&ctx->d_ois_jobs, sizeof(OisJobDev) * SVT_AMD_MAX_BATCH) != hipSuccess ||
        hipMalloc((void **)&ctx->d_jobs, sizeof(MeJobDev) * SVT_AMD_MAX_BATCH) != hipSuccess ||
        hipMalloc(&ctx->d_cabac_cost, sizeof(SvtAmdCabacCost)) != hipSuccess) {
        svt_amd_set_error("hipMalloc (context buffers) failed");
        return SVT_AMD_ERR_RESOURCES;
    }
    return SVT_AMD_OK;
}

int svt_amd_ctx_scratch(SvtAmdContext *ctx, size_t bytes, uint8_t **out)
{
    if (bytes > ctx->leaf_scratch_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->d_leaf_scratch)
            (void)hipFree(ctx->d_leaf_scratch);
        ctx->d_leaf_scratch = nullptr;
        ctx->leaf_scratch_bytes = 0;
        const size_t n = bytes < 65536 ? 65536 : bytes;
        HIP_TRY(hipMalloc((void **)&ctx->d_leaf_scratch, n));
        ctx->leaf_scratch_bytes = n;
    }
    *out = ctx->d_leaf_scratch;
    return SVT_AMD_OK;
}

/* Lanes are streams, and streams only run side by side when each has a hardware queue of its own: with the HIP runtime's default
 * of four, a fifth stream shares a queue and its markers serialise with whatever that queue holds (a result copy of one lane then
 * waits for another lane's kernels and vice versa; measured in bench.py, 1,680 -> 2,450 pictures/s).  The runtime reads the variable when it
 * starts, i.e. at the process's first HIP call - so this is a call the HOST makes, knowingly, before that (the encoder binding does, at
 * EbInitEncoder time: integration/svt_hook_me.c).  The library itself never touches the environment of the process that
 * loaded it; a setting the user made wins. */
extern "C" int svt_amd_host_wait_mode(int device_ordinal, int blocking)
{
    HIP_TRY(hipSetDevice(device_ordinal));
    HIP_TRY(hipSetDeviceFlags(blocking ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_runtime_env_defaults(void) { return setenv("GPU_MAX_HW_QUEUES", "24", 0) == 0 ? SVT_AMD_OK : SVT_AMD_ERR_RESOURCES; }

/* Job descriptors of a launch (a few KB) go host -> device through a pinned ring the GPU reads itself: a copy kernel on the
 * lane's stream, in order with the launch that consumes them.  A hipMemcpyAsync would put them on a copy engine's queue, where they
 * wait behind whatever picture-sized copy another lane has in flight - and the lane's kernels with them (measured: every batch's
 * first kernel started only when the previous batch's 680 MB result copy had finished; profiles/r02_k_timeline.txt). */
#define DESC_SLOT_BYTES (128 * 1024)
__global__ void __launch_bounds__(256) k_copy_words(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        dst[i] = src[i];
}
int svt_amd_upload_descriptors(SvtAmdContext *ctx, void *d_dst, const void *src, size_t bytes)
{
    if (bytes > DESC_SLOT_BYTES || (bytes & 3)) {
        HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return SVT_AMD_OK;
    }
    if (!ctx->h_desc_ring) {
        HIP_TRY(hipHostMalloc((void **)&ctx->h_desc_ring, (size_t)8 * DESC_SLOT_BYTES, hipHostMallocDefault));
        for (int i = 0; i < 8; i++)
            HIP_TRY(hipEventCreateWithFlags(&ctx->ev_desc[i], hipEventDisableTiming));
    }
    const int i = ctx->desc_next;
    ctx->desc_next = (i + 1) & 7;
    HIP_TRY(hipEventSynchronize(ctx->ev_desc[i])); /* the copy that last used this slot is done (8 launches ago) */
    uint8_t *slot = ctx->h_desc_ring + (size_t)i * DESC_SLOT_BYTES;
    memcpy(slot, src, bytes);
    const uint32_t n = (uint32_t)(bytes >> 2);
    hipLaunchKernelGGL(k_copy_words, dim3((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), dim3(256), 0, ctx->stream, (uint32_t *)d_dst, (const uint32_t *)slot, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_desc[i], ctx->stream));
    return SVT_AMD_OK;
}

/* A lane: a second (third, ...) stream over the SAME picture slots.  Pictures uploaded through any lane are visible to all
 * (cross-stream ordering through the slot's ev_ready event); descriptors, timers, scratch and the cost table are per lane.
 * Destroy lanes before their parent. */
extern "C" int svt_amd_context_fork(SvtAmdContext *parent, SvtAmdContext **out_lane)
{
    if (!parent || !out_lane || parent->parent) {
        svt_amd_set_error("svt_amd_context_fork: bad parameter");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    *out_lane = NULL;
    HIP_TRY(hipSetDevice(parent->device));
    SvtAmdContext *ctx = (SvtAmdContext *)calloc(1, sizeof(*ctx));
    if (!ctx)
        return SVT_AMD_ERR_RESOURCES;
    ctx->device = parent->device;
    ctx->parent = parent;
    ctx->max_w = parent->max_w;
    ctx->max_h = parent->max_h;
    ctx->num_slots = parent->num_slots;
    ctx->slots = parent->slots;
    int rc = context_common_create(ctx);
    if (rc != SVT_AMD_OK) {
        svt_amd_context_destroy(ctx);
        return rc;
    }
    *out_lane = ctx;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_context_create(int device_ordinal, uint16_t max_luma_width,
                                      uint16_t max_luma_height, int num_picture_slots,
                                      SvtAmdContext **out_ctx)
{
    if (!out_ctx || num_picture_slots < 1 || max_luma_width < 64 || max_luma_height < 64 ||
        (max_luma_width & 7) || (max_luma_height & 7)) {
        svt_amd_set_error("svt_amd_context_create: bad parameter");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    *out_ctx = NULL;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_ordinal < 0 || device_ordinal >= ndev) {
        svt_amd_set_error("svt_amd_context_create: device %d of %d", device_ordinal, ndev);
        return SVT_AMD_ERR_DEVICE;
    }
    HIP_TRY(hipSetDevice(device_ordinal));
    SvtAmdContext *ctx = (SvtAmdContext *)calloc(1, sizeof(*ctx));
    if (!ctx)
        return SVT_AMD_ERR_RESOURCES;
    ctx->device = device_ordinal;
    ctx->max_w = max_luma_width;
    ctx->max_h = max_luma_height;
    ctx->num_slots = num_picture_slots;
    ctx->slots = (DevPicture *)calloc((size_t)num_picture_slots, sizeof(DevPicture));
    int rc = ctx->slots ? SVT_AMD_OK : SVT_AMD_ERR_RESOURCES;
    const int nlcu = ((max_luma_width + 63) / 64) * ((max_luma_height + 63) / 64);
    if (rc == SVT_AMD_OK)
        rc = context_common_create(ctx);
    for (int i = 0; rc == SVT_AMD_OK && i < num_picture_slots; i++) {
        DevPicture *s = &ctx->slots[i];
        const int w = max_luma_width, h = max_luma_height;
        if ((rc = plane_create(&s->full, w, h, SVT_AMD_PAD_FULL)) != 0) break;
        if ((rc = plane_create(&s->quarter, w >> 1, h >> 1, SVT_AMD_PAD_QUARTER)) != 0) break;
        if ((rc = plane_create(&s->sixteenth, w >> 2, h >> 2, SVT_AMD_PAD_SIXTEENTH)) != 0) break;
        if ((rc = plane_create(&s->hp_b, w, h, SVT_AMD_PAD_FULL)) != 0) break;
        if ((rc = plane_create(&s->hp_h, w, h, SVT_AMD_PAD_FULL)) != 0) break;
        if ((rc = plane_create(&s->hp_j, w, h, SVT_AMD_PAD_FULL)) != 0) break;
        if (hipMalloc((void **)&s->d_me_out, (size_t)nlcu * sizeof(SvtAmdMeLcuResult)) != hipSuccess ||
            hipMalloc((void **)&s->d_ois_out, (size_t)nlcu * sizeof(SvtAmdOisLcuResult)) != hipSuccess ||
            hipMalloc(&s->d_me_carry, (size_t)nlcu * 192) != hipSuccess ||
            hipMalloc((void **)&s->d_staging, (size_t)w * h) != hipSuccess) {
            svt_amd_set_error("hipMalloc (slot %d) failed", i);
            rc = SVT_AMD_ERR_RESOURCES;
            break;
        }
        s->staging_bytes = (size_t)w * h;
        if (hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->ev_me, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_ois, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->ev_md_read, hipEventDisableTiming) != hipSuccess) {
            svt_amd_set_error("hipEventCreate (slot %d) failed", i);
            rc = SVT_AMD_ERR_DEVICE;
            break;
        }
    }
    if (rc != SVT_AMD_OK) {
        svt_amd_context_destroy(ctx);
        return rc;
    }
    *out_ctx = ctx;
    return SVT_AMD_OK;
}

/* ---- timing ------------------------------------------------------------ */

int svt_amd_stamp_begin(SvtAmdContext *ctx, int cls)
{
    if (!ctx->timer_armed)
        return SVT_AMD_OK;
    if (ctx->num_stamps == ctx->cap_stamps) {
        int ncap = ctx->cap_stamps ? ctx->cap_stamps * 2 : 64;
        SvtAmdContext::Stamp *ns = (SvtAmdContext::Stamp *)realloc(ctx->stamps, (size_t)ncap * sizeof(*ns));
        if (!ns)
            return SVT_AMD_ERR_RESOURCES;
        ctx->stamps = ns;
        for (int i = ctx->cap_stamps; i < ncap; i++) { /* cap_stamps follows every pair that exists, so none leaks on failure */
            HIP_TRY(hipEventCreate(&ns[i].a));
            if (hipEventCreate(&ns[i].b) != hipSuccess) {
                (void)hipEventDestroy(ns[i].a);
                svt_amd_set_error("hipEventCreate failed (kernel stamps)");
                return SVT_AMD_ERR_DEVICE;
            }
            ctx->cap_stamps = i + 1;
        }
    }
    ctx->stamps[ctx->num_stamps].cls = cls;
    HIP_TRY(hipEventRecord(ctx->stamps[ctx->num_stamps].a, ctx->stream));
    return SVT_AMD_OK;
}

int svt_amd_stamp_end(SvtAmdContext *ctx)
{
    if (!ctx->timer_armed)
        return SVT_AMD_OK;
    HIP_TRY(hipEventRecord(ctx->stamps[ctx->num_stamps].b, ctx->stream));
    ctx->num_stamps++;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_timer_begin(SvtAmdContext *ctx)
{
    if (!ctx)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream)); /* stamps of a previous window may still be pending */
    ctx->num_stamps = 0;
    ctx->timer_armed = 1;
    HIP_TRY(hipEventRecord(ctx->ev_begin, ctx->stream));
    return SVT_AMD_OK;
}

extern "C" int svt_amd_timer_end(SvtAmdContext *ctx, float *elapsed_ms)
{
    if (!ctx || !elapsed_ms)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipEventRecord(ctx->ev_end, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev_end));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->ev_begin, ctx->ev_end));
    ctx->timer_armed = 0;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_kernel_time(SvtAmdContext *ctx, const char *kernel_class, float *avg_ms, int *launches)
{
    if (!ctx || !kernel_class || !avg_ms || !launches)
        return SVT_AMD_ERR_BAD_PARAM;
    int cls = !strcmp(kernel_class, "prep") ? KC_PREP : !strcmp(kernel_class, "me_search") ? KC_ME_SEARCH : !strcmp(kernel_class, "ois") ? KC_OIS : -1;
    if (cls < 0)
        return SVT_AMD_ERR_BAD_PARAM;
    double sum = 0;
    int n = 0;
    for (int i = 0; i < ctx->num_stamps; i++) {
        if (ctx->stamps[i].cls != cls)
            continue;
        float ms = 0;
        HIP_TRY(hipEventSynchronize(ctx->stamps[i].b));
        HIP_TRY(hipEventElapsedTime(&ms, ctx->stamps[i].a, ctx->stamps[i].b));
        sum += ms;
        n++;
    }
    *avg_ms = n ? (float)(sum / n) : 0.f;
    *launches = n;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_synchronize(SvtAmdContext *ctx)
{
    if (!ctx)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}

/* ---- plain device memory for C hosts (reference pictures, prediction planes, job lists) ---- */
extern "C" int svt_amd_device_alloc(SvtAmdContext *ctx, size_t bytes, void **d_ptr)
{
    if (!ctx || !d_ptr || !bytes)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc(d_ptr, bytes));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_device_free(SvtAmdContext *ctx, void *d_ptr)
{
    if (!ctx)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(d_ptr));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_device_upload(SvtAmdContext *ctx, void *d_dst, const void *src, size_t bytes)
{
    if (!ctx || !d_dst || !src)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_device_copy(SvtAmdContext *ctx, void *d_dst, const void *d_src, size_t bytes)
{
    if (!ctx || !d_dst || !d_src)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_device_download(SvtAmdContext *ctx, void *dst, const void *d_src, size_t bytes)
{
    if (!ctx || !dst || !d_src)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}

/* asynchronous forms (stream-ordered on the context's stream, no host wait) and pinned host memory for them */
extern "C" int svt_amd_device_upload_async(SvtAmdContext *ctx, void *d_dst, const void *src, size_t bytes)
{
    if (!ctx || !d_dst || !src)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_device_download_async(SvtAmdContext *ctx, void *dst, const void *d_src, size_t bytes)
{
    if (!ctx || !dst || !d_src)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_host_alloc(SvtAmdContext *ctx, size_t bytes, void **h_ptr)
{
    if (!ctx || !h_ptr || !bytes)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipHostMalloc(h_ptr, bytes, hipHostMallocDefault));
    return SVT_AMD_OK;
}
/* svt_amd_host_register: the process-wide table of page-locked caller buffers (include/svt_hevc_amd.h) */
namespace {
struct HostReg { const void *p; size_t bytes; bool ok; };
std::mutex g_reg_mu;
std::vector<HostReg> g_reg;
}
extern "C" int svt_amd_host_register(SvtAmdContext *ctx, const void *h_ptr, size_t bytes)
{
    if (!ctx || !h_ptr || !bytes)
        return SVT_AMD_ERR_BAD_PARAM;
    std::lock_guard<std::mutex> g(g_reg_mu);
    for (HostReg &r : g_reg)
        if (r.p == h_ptr && r.bytes >= bytes)
            return SVT_AMD_OK;
    for (HostReg &r : g_reg) /* the same base with a larger extent, or a range overlapping an earlier one: the runtime refuses overlaps - leave it pageable */
        if ((const uint8_t *)h_ptr < (const uint8_t *)r.p + r.bytes && (const uint8_t *)r.p < (const uint8_t *)h_ptr + bytes) {
            g_reg.push_back({h_ptr, bytes, false});
            return SVT_AMD_OK;
        }
    (void)hipSetDevice(ctx->device);
    const hipError_t e = hipHostRegister(const_cast<void *>(h_ptr), bytes, hipHostRegisterDefault);
    if (e != hipSuccess)
        (void)hipGetLastError();
    g_reg.push_back({h_ptr, bytes, e == hipSuccess});
    return SVT_AMD_OK;
}
/* the device's view of [h_ptr, h_ptr + bytes) if the range lies inside a page-locked registration, nullptr otherwise */
const void *svt_amd_registered_device_ptr(const void *h_ptr, size_t bytes)
{
    std::lock_guard<std::mutex> g(g_reg_mu);
    for (const HostReg &r : g_reg)
        if (r.ok && (const uint8_t *)h_ptr >= (const uint8_t *)r.p && (const uint8_t *)h_ptr + bytes <= (const uint8_t *)r.p + r.bytes) {
            void *d = nullptr;
            if (hipHostGetDevicePointer(&d, const_cast<void *>(r.p), 0) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
            return (const uint8_t *)d + ((const uint8_t *)h_ptr - (const uint8_t *)r.p);
        }
    return nullptr;
}
extern "C" int svt_amd_host_unregister_all(SvtAmdContext *ctx)
{
    if (!ctx)
        return SVT_AMD_ERR_BAD_PARAM;
    std::lock_guard<std::mutex> g(g_reg_mu);
    (void)hipSetDevice(ctx->device);
    for (HostReg &r : g_reg)
        if (r.ok && hipHostUnregister(const_cast<void *>(r.p)) != hipSuccess)
            (void)hipGetLastError();
    g_reg.clear();
    return SVT_AMD_OK;
}
extern "C" int svt_amd_host_free(SvtAmdContext *ctx, void *h_ptr)
{
    if (!ctx)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipHostFree(h_ptr));
    return SVT_AMD_OK;
}

/* ---- pictures ---------------------------------------------------------- */

static void slot_records_reset(DevPicture *s);
static int check_slot(SvtAmdContext *ctx, int slot)
{
    if (!ctx || slot < 0 || slot >= ctx->num_slots) {
        svt_amd_set_error("bad context/slot %d", slot);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    return SVT_AMD_OK;
}

static int set_geometry(SvtAmdContext *ctx, DevPicture *s, uint16_t width, uint16_t height)
{
    if (width < 64 || height < 64 || (width & 7) || (height & 7) || width > ctx->max_w || height > ctx->max_h) {
        svt_amd_set_error("picture %ux%u not supported by context %ux%u", width, height, ctx->max_w, ctx->max_h);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    s->width = width;
    s->height = height;
    s->full.width = s->hp_b.width = s->hp_h.width = s->hp_j.width = width;
    s->full.height = s->hp_b.height = s->hp_h.height = s->hp_j.height = height;
    s->quarter.width = width >> 1, s->quarter.height = height >> 1;
    s->sixteenth.width = width >> 2, s->sixteenth.height = height >> 2;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_picture_upload_device(SvtAmdContext *ctx, int slot, const void *d_luma,
                                             uint32_t stride, uint16_t width, uint16_t height)
{
    int rc = check_slot(ctx, slot);
    if (rc)
        return rc;
    if (!d_luma || stride < width)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    DevPicture *s = &ctx->slots[slot];
    if ((rc = set_geometry(ctx, s, width, height)) != 0)
        return rc;
    if ((rc = svt_amd_launch_prep(ctx, s, (const uint8_t *)d_luma, stride)) != 0)
        return rc;
    slot_records_reset(s); /* the records in the slot's buffers are the previous picture's */
    s->valid = 1;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_picture_upload_device_batch(SvtAmdContext *ctx, int num, const int *slots,
                                                   const void *const *d_luma, uint32_t stride, uint16_t width,
                                                   uint16_t height)
{
    if (!ctx || !slots || !d_luma || num < 1 || num > SVT_AMD_MAX_BATCH || stride < width)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    DevPicture *pics[SVT_AMD_MAX_BATCH];
    for (int i = 0; i < num; i++) {
        int rc = check_slot(ctx, slots[i]);
        if (rc)
            return rc;
        if (!d_luma[i])
            return SVT_AMD_ERR_BAD_PARAM;
        pics[i] = &ctx->slots[slots[i]];
        if ((rc = set_geometry(ctx, pics[i], width, height)) != 0)
            return rc;
    }
    int rc = svt_amd_launch_prep_batch(ctx, pics, (const uint8_t *const *)d_luma, stride, num);
    if (rc)
        return rc;
    for (int i = 0; i < num; i++)
        slot_records_reset(pics[i]), pics[i]->valid = 1;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_picture_upload(SvtAmdContext *ctx, int slot, const uint8_t *luma,
                                      uint32_t stride, uint16_t width, uint16_t height)
{
    int rc = check_slot(ctx, slot);
    if (rc)
        return rc;
    if (!luma || stride < width)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    DevPicture *s = &ctx->slots[slot];
    if ((size_t)width * height > s->staging_bytes)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipMemcpy2DAsync(s->d_staging, width, luma, stride, width, height, hipMemcpyHostToDevice, ctx->stream));
    /* the caller owns `luma` and may release it as soon as we return (the reference
     * copies in EbH265EncSendPicture, EbEncHandle.c:3329): wait for the H2D copy */
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return svt_amd_picture_upload_device(ctx, slot, s->d_staging, width, width, height);
}

extern "C" int svt_amd_picture_read_plane(SvtAmdContext *ctx, int slot, int which, uint8_t *dst,
                                          size_t dst_capacity, uint32_t *out_stride,
                                          uint32_t *out_pad, uint32_t *out_rows)
{
    int rc = check_slot(ctx, slot);
    if (rc)
        return rc;
    DevPicture *s = &ctx->slots[slot];
    DevPlane *pl[6] = {&s->full, &s->quarter, &s->sixteenth, &s->hp_b, &s->hp_h, &s->hp_j};
    if (which < 0 || which > 5 || !dst || !out_stride || !out_pad || !out_rows)
        return SVT_AMD_ERR_BAD_PARAM;
    DevPlane *p = pl[which];
    const uint32_t w = (uint32_t)(p->width + 2 * p->pad), rows = (uint32_t)(p->height + 2 * p->pad);
    if (dst_capacity < (size_t)w * rows)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy2D(dst, w, p->origin - (size_t)p->pad * p->pitch - p->pad, (size_t)p->pitch, w, rows,
                        hipMemcpyDeviceToHost));
    *out_stride = w;
    *out_pad = (uint32_t)p->pad;
    *out_rows = rows;
    return SVT_AMD_OK;
}

/* ---- motion estimation ------------------------------------------------- */

static int validate_me(SvtAmdContext *ctx, const SvtAmdMeParams *p, int cur_slot, const int ref_slot[2])
{
    if (!ctx || !p || !ref_slot)
        return SVT_AMD_ERR_BAD_PARAM;
    if (p->num_lists < 1 || p->num_lists > 2 || p->num_hme_regions_w > 2 || p->num_hme_regions_h > 2 ||
        p->search_area_width < 1 || p->search_area_height < 1) {
        svt_amd_set_error("svt_amd_me_picture: bad SvtAmdMeParams");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    int rc = check_slot(ctx, cur_slot);
    for (int l = 0; !rc && l < p->num_lists; l++)
        rc = check_slot(ctx, ref_slot[l]);
    if (rc)
        return rc;
    const DevPicture *c = &ctx->slots[cur_slot];
    if (!c->valid || c->width != p->luma_width || c->height != p->luma_height) {
        svt_amd_set_error("svt_amd_me_picture: slot %d does not hold a %ux%u picture", cur_slot, p->luma_width,
                          p->luma_height);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    for (int l = 0; l < p->num_lists; l++) {
        const DevPicture *r = &ctx->slots[ref_slot[l]];
        if (!r->valid || r->width != c->width || r->height != c->height) {
            svt_amd_set_error("svt_amd_me_picture: reference slot %d invalid", ref_slot[l]);
            return SVT_AMD_ERR_BAD_PARAM;
        }
    }
    return SVT_AMD_OK;
}

/* the slot's motion-estimation / open-loop intra search records are (being) written by kernels on this lane's stream: consumers on other lanes order themselves behind */
/* A new picture enters the slot: whatever records the buffers hold belong to the previous one (every upload path calls this where it sets `valid`). */
static void slot_records_reset(DevPicture *s)
{
    __atomic_store_n(&s->me_lcus, 0u, __ATOMIC_RELEASE);
    __atomic_store_n(&s->ois_lcus, 0u, __ATOMIC_RELEASE);
    while (__atomic_exchange_n(&s->me_cov_lock, 1, __ATOMIC_ACQUIRE))
        ;
    memset(s->me_cov, 0, sizeof(s->me_cov));
    s->me_cov_count = 0;
    __atomic_store_n(&s->me_cov_lock, 0, __ATOMIC_RELEASE);
}
/* Before a kernel of this lane writes the slot's record buffers: a mode-decision kernel of another lane may still be reading the previous picture's records in place
 * (svt_amd_md_encode_picture_inter with me == NULL / ois == NULL records ev_md_read behind its launch). */
static int slot_records_before_write(SvtAmdContext *ctx, DevPicture *s)
{
    if (__atomic_load_n(&s->md_read_pending, __ATOMIC_ACQUIRE)) {
        HIP_TRY(hipStreamWaitEvent(ctx->stream, s->ev_md_read, 0));
        __atomic_store_n(&s->md_read_pending, 0, __ATOMIC_RELEASE);
    }
    return SVT_AMD_OK;
}
/* [lcu_begin, lcu_end): the LCUs the launch wrote.  The slot counts as holding the picture's records only when the launches since the last upload cover all of it. */
static int me_records_written(SvtAmdContext *ctx, int slot, const SvtAmdMeParams *p, uint32_t lcu_begin, uint32_t lcu_end)
{
    DevPicture *s = &ctx->slots[slot];
    const uint32_t n = ((p->luma_width + 63u) / 64u) * ((p->luma_height + 63u) / 64u);
    uint32_t covered;
    while (__atomic_exchange_n(&s->me_cov_lock, 1, __ATOMIC_ACQUIRE)) /* (two lanes may launch bands of one slot side by side) */
        ;
    for (uint32_t i = lcu_begin; i < lcu_end && i < n && i < 64u * 128u; i++)
        if (!((s->me_cov[i >> 6] >> (i & 63)) & 1ull))
            s->me_cov[i >> 6] |= 1ull << (i & 63), s->me_cov_count++;
    covered = s->me_cov_count;
    __atomic_store_n(&s->me_cov_lock, 0, __ATOMIC_RELEASE);
    HIP_TRY(hipEventRecord(s->ev_me, ctx->stream));
    __atomic_store_n(&s->me_lcus, covered >= n ? n : 0u, __ATOMIC_RELEASE);
    return SVT_AMD_OK;
}
static int ois_records_written(SvtAmdContext *ctx, int slot, const SvtAmdOisParams *p)
{
    DevPicture *s = &ctx->slots[slot];
    HIP_TRY(hipEventRecord(s->ev_ois, ctx->stream));
    __atomic_store_n(&s->ois_lcus, ((p->luma_width + 63u) / 64u) * ((p->luma_height + 63u) / 64u), __ATOMIC_RELEASE);
    return SVT_AMD_OK;
}

static int make_job(SvtAmdContext *ctx, const SvtAmdMeParams *params, int cur_slot, const int ref_slot[2],
                    uint32_t lcu_begin, uint32_t lcu_end, MeJobDev *job)
{
    int rc = validate_me(ctx, params, cur_slot, ref_slot);
    if (rc)
        return rc;
    const uint32_t nlcu = ((params->luma_width + 63u) / 64u) * ((params->luma_height + 63u) / 64u);
    if (lcu_begin >= lcu_end || lcu_end > nlcu) {
        svt_amd_set_error("motion estimation: bad LCU range [%u,%u) of %u", lcu_begin, lcu_end, nlcu);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    DevPicture *c = &ctx->slots[cur_slot];
    const DevPicture *r0 = &ctx->slots[ref_slot[0]];
    const DevPicture *r1 = params->num_lists == 2 ? &ctx->slots[ref_slot[1]] : r0;
    job->P = *params;
    job->cur = make_view(c);
    job->ref0 = make_view(r0);
    job->ref1 = make_view(r1);
    job->out = c->d_me_out;
    job->carry = (struct MeCarry *)c->d_me_carry;
    job->lcu_begin = (int32_t)lcu_begin;
    job->lcu_count = (int32_t)(lcu_end - lcu_begin);
    job->dbg_clock = NULL;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_me_picture_range_launch(SvtAmdContext *ctx, const SvtAmdMeParams *params, int cur_slot,
                                               const int ref_slot[2], uint32_t lcu_begin, uint32_t lcu_end)
{
    MeJobDev job;
    int rc = make_job(ctx, params, cur_slot, ref_slot, lcu_begin, lcu_end, &job);
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if ((rc = slot_records_before_write(ctx, &ctx->slots[cur_slot])) != 0)
        return rc;
    rc = svt_amd_launch_me_batch(ctx, &job, 1, job.lcu_count);
    if (rc == SVT_AMD_OK)
        rc = me_records_written(ctx, cur_slot, params, lcu_begin, lcu_end);
    return rc;
}

extern "C" int svt_amd_me_batch_launch(SvtAmdContext *ctx, const SvtAmdMeJob *jobs, int num_jobs)
{
    if (!ctx || !jobs || num_jobs < 1 || num_jobs > SVT_AMD_MAX_BATCH) {
        svt_amd_set_error("svt_amd_me_batch_launch: 1..%d jobs", SVT_AMD_MAX_BATCH);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    MeJobDev *dj = (MeJobDev *)malloc(sizeof(MeJobDev) * (size_t)num_jobs);
    if (!dj)
        return SVT_AMD_ERR_RESOURCES;
    int rc = SVT_AMD_OK, max_lcus = 0;
    for (int i = 0; i < num_jobs && rc == SVT_AMD_OK; i++) {
        const SvtAmdMeParams *p = &jobs[i].params;
        const uint32_t nlcu = ((p->luma_width + 63u) / 64u) * ((p->luma_height + 63u) / 64u);
        rc = make_job(ctx, p, jobs[i].cur_slot, jobs[i].ref_slot, 0, nlcu, &dj[i]);
        if (rc == SVT_AMD_OK)
            max_lcus = dj[i].lcu_count > max_lcus ? dj[i].lcu_count : max_lcus;
    }
    if (rc == SVT_AMD_OK) {
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess && ctx->d_dbg && (size_t)num_jobs * max_lcus <= ctx->dbg_slots)
            for (int i = 0; i < num_jobs; i++)
                dj[i].dbg_clock = ctx->d_dbg;
        for (int i = 0; i < num_jobs && e == hipSuccess && rc == SVT_AMD_OK; i++)
            rc = slot_records_before_write(ctx, &ctx->slots[jobs[i].cur_slot]);
        if (rc == SVT_AMD_OK)
            rc = e == hipSuccess ? svt_amd_launch_me_batch(ctx, dj, num_jobs, max_lcus) : SVT_AMD_ERR_DEVICE;
        for (int i = 0; i < num_jobs && rc == SVT_AMD_OK; i++)
            rc = me_records_written(ctx, jobs[i].cur_slot, &jobs[i].params, 0, (uint32_t)dj[i].lcu_count);
    }
    free(dj);
    return rc;
}

/* Phase profile of the NEXT batched ME launches: arms a device buffer of 16 shader-clock stamps
 * per workgroup; a second call with out != NULL copies the stamps of the last launch
 * (workgroups x 16 u64) and disarms.  Development aid, not part of the reference surface. */
extern "C" int svt_amd_debug_me_phase_profile(SvtAmdContext *ctx, size_t workgroups, unsigned long long *out)
{
    if (!ctx)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!out) {
        HIP_TRY(hipStreamSynchronize(ctx->stream)); /* a launch armed earlier may still write its stamps */
        if (ctx->d_dbg)
            (void)hipFree(ctx->d_dbg);
        ctx->d_dbg = NULL;
        HIP_TRY(hipMalloc((void **)&ctx->d_dbg, workgroups * 16 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(ctx->d_dbg, 0, workgroups * 16 * sizeof(unsigned long long)));
        ctx->dbg_slots = workgroups;
        return SVT_AMD_OK;
    }
    if (!ctx->d_dbg || workgroups > ctx->dbg_slots)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out, ctx->d_dbg, workgroups * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(ctx->d_dbg);
    ctx->d_dbg = NULL;
    ctx->dbg_slots = 0;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_me_picture_launch(SvtAmdContext *ctx, const SvtAmdMeParams *params, int cur_slot,
                                         const int ref_slot[2])
{
    if (!params)
        return SVT_AMD_ERR_BAD_PARAM;
    const uint32_t nlcu = ((params->luma_width + 63u) / 64u) * ((params->luma_height + 63u) / 64u);
    return svt_amd_me_picture_range_launch(ctx, params, cur_slot, ref_slot, 0, nlcu);
}

extern "C" int svt_amd_me_picture_fetch(SvtAmdContext *ctx, int cur_slot, SvtAmdMeLcuResult *out)
{
    int rc = check_slot(ctx, cur_slot);
    if (rc)
        return rc;
    if (!out)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    DevPicture *c = &ctx->slots[cur_slot];
    const int nlcu = ((c->width + 63) / 64) * ((c->height + 63) / 64);
    HIP_TRY(hipMemcpyAsync(out, c->d_me_out, (size_t)nlcu * sizeof(SvtAmdMeLcuResult), hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}

/* stream-ordered copy of the slot's ME / OIS records into (pinned) host memory; complete after svt_amd_synchronize */
extern "C" int svt_amd_me_picture_fetch_async(SvtAmdContext *ctx, int cur_slot, SvtAmdMeLcuResult *out)
{
    int rc = check_slot(ctx, cur_slot);
    if (rc)
        return rc;
    if (!out)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    DevPicture *c = &ctx->slots[cur_slot];
    const int nlcu = ((c->width + 63) / 64) * ((c->height + 63) / 64);
    HIP_TRY(hipMemcpyAsync(out, c->d_me_out, (size_t)nlcu * sizeof(SvtAmdMeLcuResult), hipMemcpyDeviceToHost, ctx->stream));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_ois_picture_fetch_async(SvtAmdContext *ctx, int cur_slot, SvtAmdOisLcuResult *out)
{
    int rc = check_slot(ctx, cur_slot);
    if (rc)
        return rc;
    if (!out)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    DevPicture *c = &ctx->slots[cur_slot];
    const int nlcu = ((c->width + 63) / 64) * ((c->height + 63) / 64);
    HIP_TRY(hipMemcpyAsync(out, c->d_ois_out, (size_t)nlcu * sizeof(SvtAmdOisLcuResult), hipMemcpyDeviceToHost, ctx->stream));
    return SVT_AMD_OK;
}

/* Cross-lane ordering for hosts that build their own pipelines (bench.py: copy-in lane -> compute lane -> copy-out lane): lane
 * `lane` records its event `index` at the current end of its stream; svt_amd_lane_event_wait makes another lane's stream wait for
 * the most recent record of it (a wait on an event that was never recorded does not wait).  Nothing blocks the host. */
extern "C" int svt_amd_lane_event_record(SvtAmdContext *lane, int index)
{
    if (!lane || index < 0 || index >= 8)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(lane->device));
    if (!lane->ev_user[index])
        HIP_TRY(hipEventCreateWithFlags(&lane->ev_user[index], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(lane->ev_user[index], lane->stream));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_lane_event_wait(SvtAmdContext *lane, SvtAmdContext *source, int index)
{
    if (!lane || !source || index < 0 || index >= 8 || lane->device != source->device)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(lane->device));
    if (source->ev_user[index])
        HIP_TRY(hipStreamWaitEvent(lane->stream, source->ev_user[index], 0));
    return SVT_AMD_OK;
}

/* ---- compact wire format of the front-half records -------------------------------------------------------------------------
 * What the host side of the boundary reads (MeCuResults_t x 85; the OIS candidates the picture's path can write: MAX_OIS_0 / _1 /
 * _2 of EbCodingUnit.h:59-61) is 5,188 of the 9,628 bytes per LCU at BASELINE configs[2]; the device packs it (one pass over the
 * records, ~10 us per 4K picture) so that the D2H copy - the longest stage of the pipelined front half - moves only that. */
__global__ void __launch_bounds__(256) k_pack_me(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int nlcu)
{
    constexpr int IN = sizeof(SvtAmdMeLcuResult) / 4, OUT = SVT_AMD_ME_PU_COUNT * sizeof(SvtAmdMeCuResult) / 4;
    const int i = blockIdx.x * 256 + threadIdx.x, lcu = i / OUT, k = i - lcu * OUT;
    if (lcu < nlcu)
        out[(size_t)lcu * OUT + k] = in[(size_t)lcu * IN + k];
}
__global__ void __launch_bounds__(256) k_pack_ois(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int nlcu, int nc)
{
    constexpr int IN = sizeof(SvtAmdOisLcuResult) / 4, TAIL = 22; /* total_intra_luma_mode[85] + pad = 88 bytes */
    const int per = SVT_AMD_ME_PU_COUNT * nc + TAIL;
    const int i = blockIdx.x * 256 + threadIdx.x, lcu = i / per, e = i - lcu * per;
    if (lcu >= nlcu)
        return;
    const int cu = e / nc, k = e - cu * nc;
    out[(size_t)lcu * per + e] =
        e < SVT_AMD_ME_PU_COUNT * nc ? in[(size_t)lcu * IN + cu * SVT_AMD_OIS_MAX_CAND + k] : in[(size_t)lcu * IN + SVT_AMD_ME_PU_COUNT * SVT_AMD_OIS_MAX_CAND + (e - SVT_AMD_ME_PU_COUNT * nc)];
}
static_assert(sizeof(SvtAmdOisLcuResult) == (SVT_AMD_ME_PU_COUNT * SVT_AMD_OIS_MAX_CAND + 22) * 4, "OIS record layout");
static_assert(sizeof(SvtAmdMeCuResult) == 24 && offsetof(SvtAmdMeLcuResult, pu) == 0, "ME record layout");

extern "C" int svt_amd_ois_compact_candidates(const SvtAmdOisParams *p)
{
    return !p ? SVT_AMD_OIS_MAX_CAND : p->slice_is_intra ? 7 : p->ois_kernel_level ? 18 : 9; /* MAX_OIS_0 / _2 / _1 */
}

static int slot_pack_buffer(SvtAmdContext *ctx, DevPicture *c, size_t bytes, uint8_t **out)
{
    if (bytes > c->pack_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (c->d_pack)
            (void)hipFree(c->d_pack);
        c->d_pack = nullptr, c->pack_bytes = 0;
        HIP_TRY(hipMalloc((void **)&c->d_pack, bytes));
        c->pack_bytes = bytes;
    }
    *out = c->d_pack;
    return SVT_AMD_OK;
}

/* the slot's pack buffer holds the ME part first, the OIS part after it (sized for the widest OIS form) */
static size_t pack_me_bytes(int nlcu) { return ((size_t)nlcu * SVT_AMD_ME_PU_COUNT * sizeof(SvtAmdMeCuResult) + 255) & ~(size_t)255; }

extern "C" int svt_amd_me_picture_fetch_compact_async(SvtAmdContext *ctx, int cur_slot, SvtAmdMeCuResult *out)
{
    int rc = check_slot(ctx, cur_slot);
    if (rc)
        return rc;
    if (!out)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    DevPicture *c = &ctx->slots[cur_slot];
    const int nlcu = ((c->width + 63) / 64) * ((c->height + 63) / 64);
    uint8_t *d = nullptr;
    if ((rc = slot_pack_buffer(ctx, c, pack_me_bytes(nlcu) + (size_t)nlcu * sizeof(SvtAmdOisLcuResult), &d)) != 0)
        return rc;
    const int n = nlcu * (int)(SVT_AMD_ME_PU_COUNT * sizeof(SvtAmdMeCuResult) / 4);
    hipLaunchKernelGGL(k_pack_me, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_me_out, (uint32_t *)d, nlcu);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    return SVT_AMD_OK;
}

extern "C" int svt_amd_ois_picture_fetch_compact_async(SvtAmdContext *ctx, int cur_slot, int candidates, void *out)
{
    int rc = check_slot(ctx, cur_slot);
    if (rc)
        return rc;
    if (!out || candidates < 1 || candidates > SVT_AMD_OIS_MAX_CAND)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    DevPicture *c = &ctx->slots[cur_slot];
    const int nlcu = ((c->width + 63) / 64) * ((c->height + 63) / 64);
    uint8_t *d = nullptr;
    if ((rc = slot_pack_buffer(ctx, c, pack_me_bytes(nlcu) + (size_t)nlcu * sizeof(SvtAmdOisLcuResult), &d)) != 0)
        return rc;
    d += pack_me_bytes(nlcu);
    const int n = nlcu * (SVT_AMD_ME_PU_COUNT * candidates + 22);
    hipLaunchKernelGGL(k_pack_ois, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_ois_out, (uint32_t *)d, nlcu, candidates);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    return SVT_AMD_OK;
}

/* A batch of pictures at once: every slot's records are packed into contiguous DEVICE arrays (picture i of the batch at index i:
 * LCUs x 85 ME records / LCUs x SVT_AMD_OIS_COMPACT_BYTES(candidates) OIS bytes).  The pack kernels belong on the lane that ran
 * the searches - small kernels on another stream starve behind its picture-sized launches - and the caller then moves each array
 * with ONE copy on whatever lane it likes (svt_amd_device_download_async).  Either destination may be NULL. */
extern "C" int svt_amd_records_pack_batch_async(SvtAmdContext *ctx, const int *slots, int n, int candidates, SvtAmdMeCuResult *d_me, void *d_ois)
{
    if (!ctx || !slots || n < 1 || n > SVT_AMD_MAX_BATCH || (!d_me && !d_ois) || candidates < 1 || candidates > SVT_AMD_OIS_MAX_CAND)
        return SVT_AMD_ERR_BAD_PARAM;
    int rc = check_slot(ctx, slots[0]);
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    const DevPicture *c0 = &ctx->slots[slots[0]];
    const int nlcu = ((c0->width + 63) / 64) * ((c0->height + 63) / 64);
    const size_t meb = (size_t)nlcu * SVT_AMD_ME_PU_COUNT * sizeof(SvtAmdMeCuResult), oisb = (size_t)nlcu * SVT_AMD_OIS_COMPACT_BYTES(candidates);
    for (int i = 0; i < n; i++) {
        if ((rc = check_slot(ctx, slots[i])) != 0)
            return rc;
        const DevPicture *c = &ctx->slots[slots[i]];
        if (c->width != c0->width || c->height != c0->height) {
            svt_amd_set_error("svt_amd_records_pack_batch_async: pictures of different sizes in one batch");
            return SVT_AMD_ERR_BAD_PARAM;
        }
        if (d_me) {
            const int k = nlcu * (int)(SVT_AMD_ME_PU_COUNT * sizeof(SvtAmdMeCuResult) / 4);
            hipLaunchKernelGGL(k_pack_me, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_me_out,
                               (uint32_t *)((uint8_t *)d_me + meb * (size_t)i), nlcu);
        }
        if (d_ois) {
            const int k = nlcu * (SVT_AMD_ME_PU_COUNT * candidates + 22);
            hipLaunchKernelGGL(k_pack_ois, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_ois_out,
                               (uint32_t *)((uint8_t *)d_ois + oisb * (size_t)i), nlcu, candidates);
        }
    }
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

extern "C" int svt_amd_me_picture(SvtAmdContext *ctx, const SvtAmdMeParams *params, int cur_slot,
                                  const int ref_slot[2], SvtAmdMeLcuResult *out)
{
    int rc = svt_amd_me_picture_launch(ctx, params, cur_slot, ref_slot);
    if (rc)
        return rc;
    return svt_amd_me_picture_fetch(ctx, cur_slot, out);
}

/* ---- open-loop intra search ------------------------------------------------ */
static int validate_ois(SvtAmdContext *ctx, const SvtAmdOisParams *p, int cur_slot)
{
    int rc = check_slot(ctx, cur_slot);
    if (rc)
        return rc;
    const DevPicture *c = &ctx->slots[cur_slot];
    if (!p || !c->valid || p->luma_width != c->width || p->luma_height != c->height || p->ois_th_set > 2 ||
        p->temporal_layer_index > 5) {
        svt_amd_set_error("svt_amd_ois_picture: bad parameter (slot %d)", cur_slot);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    return SVT_AMD_OK;
}

static void make_ois_job(SvtAmdContext *ctx, const SvtAmdOisParams *params, int cur_slot, const SvtAmdMeLcuResult *d_me,
                         OisJobDev *j)
{
    DevPicture *c = &ctx->slots[cur_slot];
    j->P = *params;
    j->full = c->full.origin;
    j->pitch = c->full.pitch;
    j->lcus_w = (params->luma_width + 63) / 64;
    j->nlcu = j->lcus_w * ((params->luma_height + 63) / 64);
    j->me = d_me;
    j->out = c->d_ois_out;
}

static int ois_launch(SvtAmdContext *ctx, const SvtAmdOisParams *params, int cur_slot, const SvtAmdMeLcuResult *d_me)
{
    HIP_TRY(hipSetDevice(ctx->device));
    OisJobDev j;
    make_ois_job(ctx, params, cur_slot, d_me, &j);
    int rc = slot_records_before_write(ctx, &ctx->slots[cur_slot]);
    if (rc || (rc = svt_amd_stamp_begin(ctx, KC_OIS)) != 0)
        return rc;
    rc = svt_amd_launch_ois_batch(ctx, &j, 1, j.nlcu);
    const int rc2 = svt_amd_stamp_end(ctx);
    if (!rc && !rc2)
        return ois_records_written(ctx, cur_slot, params);
    return rc ? rc : rc2;
}

extern "C" int svt_amd_ois_batch_launch(SvtAmdContext *ctx, const SvtAmdOisJob *jobs, int num_jobs)
{
    if (!ctx || !jobs || num_jobs < 1 || num_jobs > SVT_AMD_MAX_BATCH)
        return SVT_AMD_ERR_BAD_PARAM;
    static thread_local OisJobDev host[SVT_AMD_MAX_BATCH];
    int max_lcus = 0;
    for (int i = 0; i < num_jobs; i++) {
        int rc = validate_ois(ctx, &jobs[i].params, jobs[i].cur_slot);
        if (rc)
            return rc;
        make_ois_job(ctx, &jobs[i].params, jobs[i].cur_slot, ctx->slots[jobs[i].cur_slot].d_me_out, &host[i]);
        max_lcus = host[i].nlcu > max_lcus ? host[i].nlcu : max_lcus;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = SVT_AMD_OK;
    for (int i = 0; i < num_jobs && !rc; i++)
        rc = slot_records_before_write(ctx, &ctx->slots[jobs[i].cur_slot]);
    if (rc || (rc = svt_amd_stamp_begin(ctx, KC_OIS)) != 0)
        return rc;
    rc = svt_amd_launch_ois_batch(ctx, host, num_jobs, max_lcus);
    const int rc2 = svt_amd_stamp_end(ctx);
    for (int i = 0; i < num_jobs && !rc && !rc2; i++)
        rc = ois_records_written(ctx, jobs[i].cur_slot, &jobs[i].params);
    return rc ? rc : rc2;
}

extern "C" int svt_amd_ois_picture_launch(SvtAmdContext *ctx, const SvtAmdOisParams *params, int cur_slot)
{
    int rc = validate_ois(ctx, params, cur_slot);
    if (rc)
        return rc;
    return ois_launch(ctx, params, cur_slot, ctx->slots[cur_slot].d_me_out);
}

extern "C" int svt_amd_ois_picture_fetch(SvtAmdContext *ctx, int cur_slot, SvtAmdOisLcuResult *out)
{
    int rc = check_slot(ctx, cur_slot);
    if (rc)
        return rc;
    if (!out)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    DevPicture *c = &ctx->slots[cur_slot];
    const int nlcu = ((c->width + 63) / 64) * ((c->height + 63) / 64);
    HIP_TRY(hipMemcpyAsync(out, c->d_ois_out, (size_t)nlcu * sizeof(SvtAmdOisLcuResult), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}

extern "C" int svt_amd_ois_picture(SvtAmdContext *ctx, const SvtAmdOisParams *params, int cur_slot,
                                   const SvtAmdMeLcuResult *me, SvtAmdOisLcuResult *out)
{
    int rc = validate_ois(ctx, params, cur_slot);
    if (rc)
        return rc;
    if (!out)
        return SVT_AMD_ERR_BAD_PARAM;
    const SvtAmdMeLcuResult *d_me = ctx->slots[cur_slot].d_me_out;
    if (me) {
        HIP_TRY(hipSetDevice(ctx->device));
        const int nlcu = ((params->luma_width + 63) / 64) * ((params->luma_height + 63) / 64);
        HIP_TRY(hipMemcpyAsync(ctx->d_me_scratch, me, (size_t)nlcu * sizeof(SvtAmdMeLcuResult), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream)); /* `me` may be freed by the caller on return of a later call */
        d_me = ctx->d_me_scratch;
    }
    rc = ois_launch(ctx, params, cur_slot, d_me);
    if (rc)
        return rc;
    return svt_amd_ois_picture_fetch(ctx, cur_slot, out);
}

/* ---- front-end pipeline: asynchronous upload -> planes -> ME -> OIS -> results in pinned host memory ---------------- */

/* Copies the caller's luma into the slot's pinned staging buffer (the caller may release `luma` on return, as with
 * svt_amd_picture_upload), then queues H2D copy + plane building on this lane's stream and records the slot's ready event.
 * Nothing blocks on the device. */
extern "C" int svt_amd_picture_upload_async(SvtAmdContext *ctx, int slot, const uint8_t *luma, uint32_t stride,
                                            uint16_t width, uint16_t height)
{
    int rc = check_slot(ctx, slot);
    if (rc)
        return rc;
    if (!luma || stride < width)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    DevPicture *s = &ctx->slots[slot];
    if ((size_t)width * height > s->staging_bytes)
        return SVT_AMD_ERR_BAD_PARAM;
    if ((rc = set_geometry(ctx, s, width, height)) != 0)
        return rc;
    if (!s->h_staging)
        HIP_TRY(hipHostMalloc((void **)&s->h_staging, s->staging_bytes, hipHostMallocDefault));
    if (stride == width)
        memcpy(s->h_staging, luma, (size_t)width * height);
    else
        for (uint32_t y = 0; y < height; y++)
            memcpy(s->h_staging + (size_t)y * width, luma + (size_t)y * stride, width);
    HIP_TRY(hipMemcpyAsync(s->d_staging, s->h_staging, (size_t)width * height, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = svt_amd_launch_prep(ctx, s, s->d_staging, width)) != 0)
        return rc;
    HIP_TRY(hipEventRecord(s->ev_ready, ctx->stream));
    slot_records_reset(s); /* the records in the slot's buffers are the previous picture's */
    s->valid = 1;
    return SVT_AMD_OK;
}

/* Marks a slot filled through svt_amd_picture_upload_device[_batch] on this lane as ready for other lanes. */
extern "C" int svt_amd_picture_publish(SvtAmdContext *ctx, int slot)
{
    int rc = check_slot(ctx, slot);
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(ctx->slots[slot].ev_ready, ctx->stream));
    return SVT_AMD_OK;
}

extern "C" int svt_amd_frontend_submit(SvtAmdContext *ctx, const SvtAmdFrontendJob *job)
{
    if (!ctx || !job || (!job->has_me && !job->has_ois)) {
        svt_amd_set_error("svt_amd_frontend_submit: bad parameter");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    if (ctx->frontend_busy) {
        svt_amd_set_error("svt_amd_frontend_submit: the lane still holds an unfetched job (svt_amd_frontend_wait first)");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    int rc = check_slot(ctx, job->cur_slot);
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    DevPicture *c = &ctx->slots[job->cur_slot];
    const int nlcu = ((ctx->max_w + 63) / 64) * ((ctx->max_h + 63) / 64);
    if (!ctx->h_me)
        HIP_TRY(hipHostMalloc((void **)&ctx->h_me, (size_t)nlcu * sizeof(SvtAmdMeLcuResult), hipHostMallocDefault));
    if (!ctx->h_ois)
        HIP_TRY(hipHostMalloc((void **)&ctx->h_ois, (size_t)nlcu * sizeof(SvtAmdOisLcuResult), hipHostMallocDefault));
    /* pictures may have been prepared on another lane's stream */
    HIP_TRY(hipStreamWaitEvent(ctx->stream, c->ev_ready, 0));
    const int pn = ((c->width + 63) / 64) * ((c->height + 63) / 64);
    if (job->has_me) {
        for (int l = 0; l < job->me.num_lists && l < 2; l++) {
            if ((rc = check_slot(ctx, job->ref_slot[l])) != 0)
                return rc;
            HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->slots[job->ref_slot[l]].ev_ready, 0));
        }
        if ((rc = svt_amd_me_picture_launch(ctx, &job->me, job->cur_slot, job->ref_slot)) != 0)
            return rc;
        if (job->compact)
            rc = svt_amd_me_picture_fetch_compact_async(ctx, job->cur_slot, (SvtAmdMeCuResult *)ctx->h_me);
        else
            HIP_TRY(hipMemcpyAsync(ctx->h_me, c->d_me_out, (size_t)pn * sizeof(SvtAmdMeLcuResult), hipMemcpyDeviceToHost, ctx->stream));
        if (rc)
            return rc;
    }
    if (job->has_ois) {
        if ((rc = svt_amd_ois_picture_launch(ctx, &job->ois, job->cur_slot)) != 0)
            return rc;
        if (job->compact)
            rc = svt_amd_ois_picture_fetch_compact_async(ctx, job->cur_slot, svt_amd_ois_compact_candidates(&job->ois), ctx->h_ois);
        else
            HIP_TRY(hipMemcpyAsync(ctx->h_ois, c->d_ois_out, (size_t)pn * sizeof(SvtAmdOisLcuResult), hipMemcpyDeviceToHost, ctx->stream));
        if (rc)
            return rc;
    }
    HIP_TRY(hipEventRecord(ctx->ev_done, ctx->stream));
    ctx->frontend_busy = 1;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_frontend_wait(SvtAmdContext *ctx, const SvtAmdMeLcuResult **me, const SvtAmdOisLcuResult **ois)
{
    if (!ctx || !ctx->frontend_busy) {
        svt_amd_set_error("svt_amd_frontend_wait: nothing submitted on this lane");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(ctx->ev_done));
    if (me)
        *me = ctx->h_me;
    if (ois)
        *ois = ctx->h_ois;
    return SVT_AMD_OK;
}

/* The lane's pinned result buffers may be overwritten by the next submit from now on. */
extern "C" int svt_amd_frontend_release(SvtAmdContext *ctx)
{
    if (!ctx)
        return SVT_AMD_ERR_BAD_PARAM;
    ctx->frontend_busy = 0;
    return SVT_AMD_OK;
}

/* Everything a first picture would otherwise pay for inside the encoder's timed run: the pinned staging buffer of every picture
 * slot, the lane's pinned result buffers, and the code objects of the front-half kernels (loaded on first launch) - one dummy
 * picture goes through upload -> planes -> ME -> OIS on this lane and is dropped again.  Called per lane at encoder start-up. */
extern "C" int svt_amd_frontend_warmup(SvtAmdContext *ctx)
{
    if (!ctx)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    const int nlcu = ((ctx->max_w + 63) / 64) * ((ctx->max_h + 63) / 64);
    if (!ctx->h_me)
        HIP_TRY(hipHostMalloc((void **)&ctx->h_me, (size_t)nlcu * sizeof(SvtAmdMeLcuResult), hipHostMallocDefault));
    if (!ctx->h_ois)
        HIP_TRY(hipHostMalloc((void **)&ctx->h_ois, (size_t)nlcu * sizeof(SvtAmdOisLcuResult), hipHostMallocDefault));
    if (!ctx->parent) { /* the root owns the slots */
        for (int i = 0; i < ctx->num_slots; i++) {
            DevPicture *s = &ctx->slots[i];
            if (!s->h_staging)
                HIP_TRY(hipHostMalloc((void **)&s->h_staging, s->staging_bytes, hipHostMallocDefault));
            uint8_t *d = nullptr;
            int rcp = slot_pack_buffer(ctx, s, pack_me_bytes(nlcu) + (size_t)nlcu * sizeof(SvtAmdOisLcuResult), &d);
            if (rcp)
                return rcp;
        }
    }
    if (ctx->frontend_busy || ctx->num_slots < 1)
        return SVT_AMD_OK;
    /* a mid-grey picture against itself */
    const uint16_t w = ctx->max_w, h = ctx->max_h;
    DevPicture *s0 = &ctx->slots[0];
    if (s0->valid)
        return SVT_AMD_OK; /* slot 0 is in use: the device is warm already */
    if (!s0->h_staging)
        HIP_TRY(hipHostMalloc((void **)&s0->h_staging, s0->staging_bytes, hipHostMallocDefault));
    std::vector<uint8_t> grey((size_t)w * h, 128);
    int rc = svt_amd_picture_upload_async(ctx, 0, grey.data(), w, w, h);
    if (rc)
        return rc;
    SvtAmdFrontendJob job;
    memset(&job, 0, sizeof(job));
    job.cur_slot = 0, job.ref_slot[0] = job.ref_slot[1] = 0;
    job.has_me = 1, job.has_ois = 1;
    SvtAmdMeParams &p = job.me;
    p.luma_width = w, p.luma_height = h, p.num_lists = 2;
    p.enable_hme_flag = p.enable_hme_level0 = p.enable_hme_level1 = 1;
    p.update_hme_search_center = 1, p.num_hme_regions_w = p.num_hme_regions_h = 2;
    p.search_area_width = 16, p.search_area_height = 9, p.fractional_search_model = 1, p.cu8x8_mode = 1;
    p.hme_l0_total_w = 64, p.hme_l0_total_h = 32;
    for (int k = 0; k < 2; k++) {
        p.hme_l0_w[k] = 32, p.hme_l0_h[k] = 16, p.hme_l1_w[k] = 8, p.hme_l1_h[k] = 8, p.hme_l2_w[k] = 4, p.hme_l2_h[k] = 4;
    }
    p.hme_l0_mult_x = p.hme_l0_mult_y = 100, p.lambda = 200;
    for (int k = 0; k < 10; k++)
        p.mvd_bits[k] = 16384u + 4096u * (uint32_t)k;
    SvtAmdOisParams &o = job.ois;
    o.luma_width = w, o.luma_height = h, o.ois_th_set = 1, o.cu8x8_mode = 1;
    if ((rc = svt_amd_frontend_submit(ctx, &job)) != 0)
        return rc;
    const SvtAmdMeLcuResult *me;
    const SvtAmdOisLcuResult *ois;
    if ((rc = svt_amd_frontend_wait(ctx, &me, &ois)) != 0)
        return rc;
    svt_amd_frontend_release(ctx);
    s0->valid = 0;
    return SVT_AMD_OK;
}

/* ---- collocated zero-motion SAD -------------------------------------------- */
extern "C" int svt_amd_zz_sad_picture(SvtAmdContext *ctx, int cur_slot, int prev_slot, SvtAmdZzLcu *out)
{
    int rc = check_slot(ctx, cur_slot);
    if (!rc)
        rc = check_slot(ctx, prev_slot);
    if (rc)
        return rc;
    DevPicture *c = &ctx->slots[cur_slot], *p = &ctx->slots[prev_slot];
    if (!out || !c->valid || !p->valid || c->width != p->width || c->height != p->height) {
        svt_amd_set_error("svt_amd_zz_sad_picture: bad parameter (slots %d, %d)", cur_slot, prev_slot);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const int nlcu = ((c->width + 63) / 64) * ((c->height + 63) / 64);
    /* results are small (8 B / LCU): reuse the head of the ME scratch buffer */
    SvtAmdZzLcu *d_out = (SvtAmdZzLcu *)ctx->d_me_scratch;
    if ((rc = svt_amd_launch_zz_sad(ctx, c, p, d_out)) != 0)
        return rc;
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)nlcu * sizeof(SvtAmdZzLcu), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}
