/*
 * Multi-GPU exchange of the closed-loop half (SURVEY.md 8e, row "EncDec with tiles"): tiles of one picture are independent
 * inside the picture (loop_filter_across_tiles = 0, Codec/EbEntropyCoding.c:6346-6351; per-tile neighbour arrays,
 * Codec/EbEncDecProcess.c:2743-2760), so rank r encodes its tiles and owns their part of the reconstructed picture.  Motion
 * vectors of LATER pictures may cross tile borders (unrestrictedMotionVector = 1, EbEncHandle.c:2757), so once a reference
 * picture is finished (after DLF / SAO / padding: PadRefAndSetFlags, EbEncDecProcess.c:1806) every rank needs all of it:
 * ONE all-gather per reference picture over xGMI - the only data-path collective of the design.
 *
 *   svt_amd_tile_partition   the reference's uniform tile grid (tileColStartLcu[c] = c * widthInLcu / cols,
 *                            Codec/EbPictureControlSet.c:743-750) and the rectangle of tiles each rank owns (host code)
 *   svt_amd_comm_*           RCCL communicator owned by a context; librccl is opened at run time (dlopen), so the library
 *                            loads - and single-GPU use works - on hosts without it
 *   svt_amd_recon_exchange   pack own rectangle (Y, Cb, Cr) -> ncclAllGather -> unpack everybody else's rectangles into
 *                            the local planes.  Bytes on the wire per rank: (G-1) x the largest rectangle (4:2:0: 1.5 x w x h x
 *                            bytes per sample); 8K 10-bit, 8 ranks: 99.5 MB per picture in total.
 */
#include <dlfcn.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include "svt_amd_internal.h"

/* ---- tile grid ------------------------------------------------------------------------------------------------------ */
extern "C" int svt_amd_tile_partition(uint16_t luma_width, uint16_t luma_height, int tile_cols, int tile_rows, int world,
                                      SvtAmdRect *rank_rect, int *tile_rank)
{
    if (luma_width < 64 || luma_height < 64 || tile_cols < 1 || tile_rows < 1 || world < 1 || !rank_rect) {
        svt_amd_set_error("svt_amd_tile_partition: bad parameter");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    const int wl = (luma_width + 63) / 64, hl = (luma_height + 63) / 64;
    if (tile_cols > wl || tile_rows > hl) {
        svt_amd_set_error("svt_amd_tile_partition: %d x %d tiles on %d x %d LCUs", tile_cols, tile_rows, wl, hl);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    /* ranks form a grid over the tile grid: as many rank columns as divide the tile columns, the rest over tile rows, so that
     * every rank owns a RECTANGLE of whole tiles (one strided copy per plane).  cfg5: 4 tile columns, 8 ranks -> 4 x 2. */
    int rc = 1;
    for (int d = 1; d <= world; d++)
        if (world % d == 0 && tile_cols % d == 0)
            rc = d;
    const int rr = world / rc;
    if (tile_rows % rr != 0) {
        svt_amd_set_error("svt_amd_tile_partition: %d ranks do not tile a %d x %d tile grid (rank grid %d x %d)", world, tile_cols,
                          tile_rows, rc, rr);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    const int tpc = tile_cols / rc, tpr = tile_rows / rr; /* tiles per rank, per direction */
    for (int r = 0; r < world; r++) {
        const int gx = r % rc, gy = r / rc;
        const int c0 = gx * tpc, c1 = c0 + tpc, r0 = gy * tpr, r1 = r0 + tpr;
        const int x0 = (c0 * wl / tile_cols) * 64, x1 = c1 == tile_cols ? luma_width : (c1 * wl / tile_cols) * 64;
        const int y0 = (r0 * hl / tile_rows) * 64, y1 = r1 == tile_rows ? luma_height : (r1 * hl / tile_rows) * 64;
        rank_rect[r].x = (uint16_t)x0, rank_rect[r].y = (uint16_t)y0;
        rank_rect[r].w = (uint16_t)(x1 - x0), rank_rect[r].h = (uint16_t)(y1 - y0);
    }
    if (tile_rank)
        for (int ty = 0; ty < tile_rows; ty++)
            for (int tx = 0; tx < tile_cols; tx++)
                tile_rank[ty * tile_cols + tx] = (ty / tpr) * rc + tx / tpc;
    return SVT_AMD_OK;
}

/* ---- RCCL, opened at run time ------------------------------------------------------------------------------------------ */
typedef struct { char internal[128]; } RcclUniqueId; /* ncclUniqueId, rccl.h:43 */
typedef void *RcclComm;
struct Rccl {
    void *so;
    int (*GetUniqueId)(RcclUniqueId *);
    int (*CommInitRank)(RcclComm *, int, RcclUniqueId, int);
    int (*CommDestroy)(RcclComm);
    int (*AllGather)(const void *, void *, size_t, int, RcclComm, hipStream_t);
    int (*Broadcast)(const void *, void *, size_t, int, int, RcclComm, hipStream_t);
    const char *(*GetErrorString)(int);
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;

static int rccl_open(void)
{
    std::lock_guard<std::mutex> g(g_rccl_mu);
    if (g_rccl.so)
        return SVT_AMD_OK;
    const char *names[] = {getenv("SVT_AMD_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *so = NULL;
    for (unsigned i = 0; i < sizeof(names) / sizeof(names[0]) && !so; i++)
        if (names[i])
            so = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!so) {
        svt_amd_set_error("RCCL not found (librccl.so.1): %s", dlerror());
        return SVT_AMD_ERR_DEVICE;
    }
    Rccl r;
    r.so = so;
    *(void **)&r.GetUniqueId = dlsym(so, "ncclGetUniqueId");
    *(void **)&r.CommInitRank = dlsym(so, "ncclCommInitRank");
    *(void **)&r.CommDestroy = dlsym(so, "ncclCommDestroy");
    *(void **)&r.AllGather = dlsym(so, "ncclAllGather");
    *(void **)&r.Broadcast = dlsym(so, "ncclBroadcast");
    *(void **)&r.GetErrorString = dlsym(so, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Broadcast) {
        svt_amd_set_error("RCCL: missing symbols");
        return SVT_AMD_ERR_DEVICE;
    }
    g_rccl = r;
    return SVT_AMD_OK;
}
#define RCCL_TRY(call)                                                                                              \
    do {                                                                                                            \
        const int r_ = (call);                                                                                      \
        if (r_ != 0) {                                                                                              \
            svt_amd_set_error("%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"); \
            return SVT_AMD_ERR_DEVICE;                                                                              \
        }                                                                                                           \
    } while (0)

extern "C" int svt_amd_comm_unique_id(SvtAmdCommId *out)
{
    if (!out)
        return SVT_AMD_ERR_BAD_PARAM;
    int rc = rccl_open();
    if (rc)
        return rc;
    static_assert(sizeof(SvtAmdCommId) == sizeof(RcclUniqueId), "id size");
    RCCL_TRY(g_rccl.GetUniqueId((RcclUniqueId *)out));
    return SVT_AMD_OK;
}

extern "C" int svt_amd_comm_init(SvtAmdContext *ctx, int world, int rank, const SvtAmdCommId *id)
{
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world || ctx->comm)
        return SVT_AMD_ERR_BAD_PARAM;
    int rc = rccl_open();
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    RcclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    RCCL_TRY(g_rccl.CommInitRank((RcclComm *)&ctx->comm, world, uid, rank));
    ctx->comm_world = world, ctx->comm_rank = rank;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_comm_destroy(SvtAmdContext *ctx)
{
    if (!ctx)
        return SVT_AMD_ERR_BAD_PARAM;
    if (ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        g_rccl.CommDestroy((RcclComm)ctx->comm);
        ctx->comm = NULL;
    }
    if (ctx->d_xchg)
        (void)hipFree(ctx->d_xchg);
    ctx->d_xchg = NULL, ctx->xchg_bytes = 0;
    return SVT_AMD_OK;
}

/* ---- pack / unpack: rectangles of the three planes <-> one contiguous slot per rank ---------------------------------------
 * slot layout: Y rows (w * bps bytes each, h rows), then Cb, then Cr (w/2 x h/2).  One thread moves 16 bytes. */
struct XchgPlanes {
    uint8_t *p[3];
    uint32_t pitch[3]; /* bytes */
};
__global__ __launch_bounds__(256) void k_xchg_copy(XchgPlanes P, uint8_t *__restrict__ slots, size_t slot_bytes, const SvtAmdRect *__restrict__ rects,
                                                   int bps, int first, int count, int skip, int to_slot)
{
    /* blockIdx.y = rectangle (rank), blockIdx.x * 256 + t = 16-byte item of that rectangle */
    const int ri = first + (int)blockIdx.y;
    if (ri >= first + count || ri == skip)
        return;
    const SvtAmdRect R = rects[ri];
    const uint32_t wb = (uint32_t)R.w * bps, cwb = wb >> 1; /* row bytes luma / chroma */
    const uint32_t items_y = ((wb + 15) >> 4) * R.h, items_c = ((cwb + 15) >> 4) * (R.h >> 1);
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    int plane = 0;
    if (i >= items_y) {
        i -= items_y, plane = 1;
        if (i >= items_c)
            i -= items_c, plane = 2;
        if (i >= items_c)
            return;
    }
    const uint32_t rowb = plane ? cwb : wb, per_row = (rowb + 15) >> 4;
    const uint32_t row = i / per_row, col = (i - row * per_row) << 4;
    const uint32_t n = rowb - col < 16 ? rowb - col : 16;
    uint8_t *pl = P.p[plane] + (size_t)((plane ? R.y >> 1 : R.y) + row) * P.pitch[plane] + (size_t)(plane ? R.x >> 1 : R.x) * bps + col;
    uint8_t *sl = slots + (size_t)ri * slot_bytes + (plane == 0 ? 0 : (size_t)wb * R.h + (plane == 2 ? (size_t)cwb * (R.h >> 1) : 0)) +
                  (size_t)row * rowb + col;
    uint8_t *dst = to_slot ? sl : pl;
    const uint8_t *src = to_slot ? pl : sl;
    if (n == 16 && !(((uintptr_t)dst | (uintptr_t)src) & 15))
        *(uint4 *)dst = *(const uint4 *)src;
    else
        for (uint32_t k = 0; k < n; k++)
            dst[k] = src[k];
}

static size_t rect_bytes(const SvtAmdRect &r, int bps) { return (size_t)r.w * r.h * bps * 3 / 2; }

static int xchg_prepare(SvtAmdContext *ctx, const SvtAmdRect *rects, int world, int bps, size_t *slot_bytes)
{
    size_t mx = 0;
    for (int r = 0; r < world; r++) {
        if ((rects[r].w & 1) || (rects[r].h & 1) || (rects[r].x & 1) || (rects[r].y & 1) || !rects[r].w || !rects[r].h) {
            svt_amd_set_error("recon exchange: rectangle %d is not even-aligned", r);
            return SVT_AMD_ERR_BAD_PARAM;
        }
        const size_t b = rect_bytes(rects[r], bps);
        mx = b > mx ? b : mx;
    }
    mx = (mx + 255) & ~(size_t)255;
    const size_t need = mx * (size_t)world + (size_t)world * sizeof(SvtAmdRect) + 256;
    if (need > ctx->xchg_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->d_xchg)
            (void)hipFree(ctx->d_xchg);
        ctx->d_xchg = NULL, ctx->xchg_bytes = 0;
        HIP_TRY(hipMalloc((void **)&ctx->d_xchg, need));
        ctx->xchg_bytes = need;
    }
    *slot_bytes = mx;
    return SVT_AMD_OK;
}

/* what one rank does per finished reference picture; rects[world] = svt_amd_tile_partition's rectangles; the planes are DEVICE
 * pointers to sample (0,0) of Y / Cb / Cr of the local reconstructed picture, pitches in BYTES.  Stream-ordered on the context's
 * stream; with a communicator of size 1 (or none and world == 1) it degenerates to nothing. */
extern "C" int svt_amd_recon_exchange(SvtAmdContext *ctx, void *const d_planes[3], const uint32_t pitch_bytes[3], int bytes_per_sample,
                                      const SvtAmdRect *rects, int world, int rank)
{
    if (!ctx || !d_planes || !pitch_bytes || !rects || world < 1 || rank < 0 || rank >= world || (bytes_per_sample != 1 && bytes_per_sample != 2))
        return SVT_AMD_ERR_BAD_PARAM;
    if (world == 1)
        return SVT_AMD_OK;
    if (!ctx->comm || ctx->comm_world != world || ctx->comm_rank != rank) {
        svt_amd_set_error("svt_amd_recon_exchange: no communicator of %d ranks on this context (svt_amd_comm_init)", world);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    size_t slot = 0;
    int rc = xchg_prepare(ctx, rects, world, bytes_per_sample, &slot);
    if (rc)
        return rc;
    SvtAmdRect *d_rects = (SvtAmdRect *)(ctx->d_xchg + slot * (size_t)world);
    HIP_TRY(hipMemcpyAsync(d_rects, rects, sizeof(SvtAmdRect) * (size_t)world, hipMemcpyHostToDevice, ctx->stream));
    XchgPlanes P;
    for (int p = 0; p < 3; p++)
        P.p[p] = (uint8_t *)d_planes[p], P.pitch[p] = pitch_bytes[p];
    size_t mx_items = 0;
    for (int r = 0; r < world; r++) {
        const size_t wb = (size_t)rects[r].w * bytes_per_sample;
        const size_t it = ((wb + 15) >> 4) * rects[r].h + 2 * (((wb / 2) + 15) >> 4) * (rects[r].h >> 1);
        mx_items = it > mx_items ? it : mx_items;
    }
    const unsigned gx = (unsigned)((mx_items + 255) / 256);
    /* own rectangle -> own slot (in place all-gather: send buffer = own slot of the receive buffer) */
    hipLaunchKernelGGL(k_xchg_copy, dim3(gx, 1), dim3(256), 0, ctx->stream, P, ctx->d_xchg, slot, d_rects, bytes_per_sample, rank, 1, -1, 1);
    RCCL_TRY(g_rccl.AllGather(ctx->d_xchg + slot * (size_t)rank, ctx->d_xchg, slot, 0 /* ncclInt8 */, (RcclComm)ctx->comm, ctx->stream));
    /* everybody else's slots -> the local planes */
    hipLaunchKernelGGL(k_xchg_copy, dim3(gx, (unsigned)world), dim3(256), 0, ctx->stream, P, ctx->d_xchg, slot, d_rects, bytes_per_sample, 0, world,
                       rank, 0);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* Picture-level parallelism (SURVEY 8e last row): a picture is encoded WHOLE by the rank that owns it; the only traffic is the finished reference picture, sent by its
 * owner once - ncclBroadcast of the three planes (whole allocations of `bytes[p]` bytes, the same on every rank), stream-ordered on the context's stream. */
extern "C" int svt_amd_recon_broadcast(SvtAmdContext *ctx, void *const d_planes[3], const size_t bytes[3], int world, int rank, int root)
{
    if (!ctx || !d_planes || !bytes || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world)
        return SVT_AMD_ERR_BAD_PARAM;
    if (world == 1)
        return SVT_AMD_OK;
    if (!ctx->comm || ctx->comm_world != world || ctx->comm_rank != rank) {
        svt_amd_set_error("svt_amd_recon_broadcast: no communicator of %d ranks on this context (svt_amd_comm_init)", world);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    for (int p = 0; p < 3; p++)
        RCCL_TRY(g_rccl.Broadcast(d_planes[p], d_planes[p], bytes[p], 0 /* ncclInt8 */, root, (RcclComm)ctx->comm, ctx->stream));
    return SVT_AMD_OK;
}

/* The two local halves of the exchange without a communicator: rectangle `r` of the planes -> slot r of d_slots (to_slot) or back.
 * What tests use to check the packing on one GPU, and what a host that moves the slots itself (e.g. over its own transport) calls. */
extern "C" int svt_amd_recon_pack(SvtAmdContext *ctx, void *const d_planes[3], const uint32_t pitch_bytes[3], int bytes_per_sample,
                                  const SvtAmdRect *rects, int world, int r, void *d_slots, size_t slot_bytes, int to_slot)
{
    if (!ctx || !d_planes || !pitch_bytes || !rects || !d_slots || r < 0 || r >= world || (bytes_per_sample != 1 && bytes_per_sample != 2) ||
        slot_bytes < rect_bytes(rects[r], bytes_per_sample))
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    size_t slot = 0;
    int rc = xchg_prepare(ctx, rects, world, bytes_per_sample, &slot);
    if (rc)
        return rc;
    SvtAmdRect *d_rects = (SvtAmdRect *)(ctx->d_xchg + slot * (size_t)world);
    HIP_TRY(hipMemcpyAsync(d_rects, rects, sizeof(SvtAmdRect) * (size_t)world, hipMemcpyHostToDevice, ctx->stream));
    XchgPlanes P;
    for (int p = 0; p < 3; p++)
        P.p[p] = (uint8_t *)d_planes[p], P.pitch[p] = pitch_bytes[p];
    const size_t wb = (size_t)rects[r].w * bytes_per_sample;
    const size_t items = ((wb + 15) >> 4) * rects[r].h + 2 * (((wb / 2) + 15) >> 4) * (rects[r].h >> 1);
    hipLaunchKernelGGL(k_xchg_copy, dim3((unsigned)((items + 255) / 256), 1), dim3(256), 0, ctx->stream, P, (uint8_t *)d_slots, slot_bytes, d_rects,
                       bytes_per_sample, r, 1, -1, to_slot);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
