/*
 * Device-resident encode pass: the batched boundary `hip_encdec_segment` of SURVEY.md 8(b) for the final encode pass.
 *
 * Replaces, per LCU, the coding-unit loop of EncodePass (Codec/EbCodingLoop.c:2989, CU loop :3180-4594) for intra coding units:
 *     GenerateIntraReferenceSamplesEncodePass + EncodePassIntraPrediction   (:3292-3352; Codec/EbIntraPrediction.c:212, 4395)
 *     EncodeLoop      PictureResidual -> EstimateTransform -> UnifiedQuantizeInvQuantize, Y / Cb / Cr   (:651-1080)
 *     EncodeGenerateRecon   EncodeInvTransform (DC-only shortcut) + PictureAdditionKernel               (:1084-1243)
 *     EncodePassUpdateReconSampleNeighborArrays / ...IntraModeNeighborArrays                              (:3437-3520)
 * ONE launch serves every LCU the host declares ready (the LCUs of one wavefront step of AssignEncDecSegments,
 * Codec/EbEncDecProcess.c:1540): one 256-thread workgroup per LCU walks the LCU's final coding-unit list in order - the closed loop
 * inside an LCU (a unit predicts from the reconstruction of the units before it) runs on the device, nothing returns to the host
 * between units.
 *
 * Device-resident state of a picture (SvtAmdEncDecPicture):
 *   - the reconstruction BEFORE deblocking, three planes: the reference keeps only the last row / column of every unit in its
 *     "neighbour arrays" (epLumaReconNeighborArray ..., Codec/EbNeighborArrays.c:113) because its picture buffer is deblocked in
 *     place; a unit's left / top / top-left neighbours are always on the bottom row or right column of the unit that holds them and
 *     that unit is the last writer of the array entry (Z-order is monotone in x and y), so reading the un-deblocked picture at
 *     (x0 - 1, y), (x, y0 - 1), (x0 - 1, y0 - 1) returns exactly the array contents wherever the reference may read them;
 *   - the mode-type map, one byte per 4x4 luma block (0xFF not coded yet, 1 INTER, 2 INTRA) = epModeTypeNeighborArray with the same
 *     last-writer argument; it decides the availability of every 4-sample neighbour group (constrained intra included).
 * Per LCU the host sends the coding-unit list + the source samples (SvtAmdLcuWork) and gets back the quantised coefficients in the
 * layout of LargestCodingUnit_t.quantizedCoeff, the cbf / DC-only / count fields of every TransformUnit_t and the un-deblocked
 * reconstruction of the LCU (SvtAmdLcuResult): the EncDec output contract of SURVEY 8(a) for the encode pass.
 *
 * Transform unit on N lanes of one wave (Y on wave 0, Cb on wave 1, Cr on wave 2, concurrently): row r of source and prediction ->
 * residual -> forward "Estimate" DCT in registers -> column r quantised / de-quantised in registers -> inverse DCT -> + prediction.
 */
#include "txfm_device.h"
#include <string.h>
#include <vector>
#include <mutex>
#include "intra_device.h"
#include "rate_device.h"
#include "pmcore_device.h"

#include "encdec_device.h"

template <typename T>
__global__ __launch_bounds__(256) void k_encode_lcu(EpPicture P, const typename EpTypes<T>::Work *__restrict__ works,
                                                    typename EpTypes<T>::Result *__restrict__ results)
{
    __shared__ EpShared<T> S;
    __shared__ EpLocal<T> L;
    ep_encode_lcu<T>(P, works[blockIdx.x], results[blockIdx.x], S, L);
}

/* AssignEncDecSegments on the device (Codec/EbEncDecProcess.c:1540: an LCU may start when its left and its top-right LCUs are done):
 * ONE launch encodes a whole picture.  A small persistent grid of workgroups draws LCUs as tickets in raster order - every LCU an
 * LCU waits for has a lower ticket, so it is finished or held by a workgroup that is running, and the wait cannot deadlock.  An LCU
 * publishes itself with a device-scope release (its samples, mode map cells and results are then visible to every XCD's L2); a
 * (tickets follow the wavefront's anti-diagonals, so a grid as wide as the wavefront stays busy).  A
 * waiting workgroup polls the flags with RELAXED loads (an acquire per poll would drop the XCD's L2 contents every few hundred
 * cycles and starve the workgroups that do the work - measured: 20x slower) and acquires ONCE before it reads its neighbours.
 * done[] holds the epoch of the call that finished the LCU.
 * WHAT an LCU waits for is what its units read of their neighbours: only intra units read reconstructed samples and mode types across the LCU
 * border (left, top-left, top and top-right LCU of the same tile); inter units predict from the reference pictures.  An LCU without intra units
 * therefore starts at once - in a P / B picture most do, and the wavefront of AssignEncDecSegments (which the reference needs for its
 * neighbour ARRAYS) degenerates into nearly independent LCUs. */
template <typename T>
__global__ __launch_bounds__(256) void k_encode_picture(EpPicture P, const typename EpTypes<T>::Work *__restrict__ works,
                                                        typename EpTypes<T>::Result *__restrict__ results, int nlcu,
                                                        int wl, unsigned *ticket, unsigned *done, const unsigned *__restrict__ order, unsigned epoch)
{
    __shared__ EpShared<T> S;
    __shared__ EpLocal<T> L;
    __shared__ unsigned s_ticket;
    for (;;) {
        __syncthreads(); /* the previous LCU's readers of s_ticket are through */
        if (threadIdx.x == 0)
            s_ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        if ((int)s_ticket >= nlcu)
            return;
        const int lcu = (int)order[s_ticket];
        const typename EpTypes<T>::Work &W = works[lcu];
        const unsigned long long w0 = P.prof ? __builtin_readcyclecounter() : 0;
        if (threadIdx.x == 0)
            ep_wait_neighbours(W, lcu, wl, done, epoch);
        __syncthreads();
        const unsigned long long w1 = P.prof ? __builtin_readcyclecounter() : 0;
        ep_encode_lcu<T>(P, W, results[lcu], S, L);
        __syncthreads(); /* every thread's stores of this LCU are issued */
        if (P.prof && threadIdx.x == 0)
            P.prof[16 * (size_t)lcu + 4] = w1 - w0, P.prof[16 * (size_t)lcu + 5] = w0, P.prof[16 * (size_t)lcu + 6] = __builtin_readcyclecounter();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(&done[lcu], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

/* ---- host side ------------------------------------------------------------------------------------------------------------- */
/* the encode pass behind the mode-decision kernel of the same picture-level call (md_kernels.hip): d_works / d_results are the picture object's device arrays of the call's
 * sample width; a P / B picture - most LCUs without an intra unit wait for nobody - as wide as the device holds, an I picture as wide as its wavefront */
int svt_amd_ep_launch_behind_md(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const void *d_works, void *d_results, int n_active, const unsigned *d_order, int inter, int tiles)
{
    const int wl = (pic->d.width + 63) / 64, hl = (pic->d.height + 63) / 64;
    int grid = inter ? 512 : ((wl + 1) / 2 < hl ? (wl + 1) / 2 : hl) * (tiles > 0 ? tiles : 1) + 1;
    grid = grid > n_active ? n_active : grid;
    if (pic->d.bps == 2)
        hipLaunchKernelGGL(k_encode_picture<uint16_t>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, pic->d, (const SvtAmdLcuWork16 *)d_works, (SvtAmdLcuResult16 *)d_results, n_active, wl,
                           pic->d_sync, pic->d_sync + 1, d_order, pic->epoch);
    else
        hipLaunchKernelGGL(k_encode_picture<uint8_t>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, pic->d, (const SvtAmdLcuWork *)d_works, (SvtAmdLcuResult *)d_results, n_active, wl,
                           pic->d_sync, pic->d_sync + 1, d_order, pic->epoch);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

static int picture_create(SvtAmdContext *ctx, uint16_t width, uint16_t height, int bytes_per_sample, SvtAmdEncDecPicture **out)
{
    if (!ctx || !out || width < 8 || height < 8 || (width & 7) || (height & 7) || (bytes_per_sample != 1 && bytes_per_sample != 2)) {
        svt_amd_set_error("svt_amd_encdec_picture_create: bad parameter (1 or 2 bytes per sample, dimensions multiples of 8)");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    SvtAmdEncDecPicture *p = (SvtAmdEncDecPicture *)calloc(1, sizeof(*p));
    if (!p)
        return SVT_AMD_ERR_RESOURCES;
    *out = p; /* published at once: the caller releases it when a later step fails */
    p->device = ctx->device;
    p->d.width = width, p->d.height = height, p->d.bps = (uint32_t)bytes_per_sample;
    for (int k = 0; k < 3; k++) {
        const uint32_t w = k ? width >> 1 : width, h = k ? height >> 1 : height;
        p->d.pitch[k] = (w + 127) & ~127u;
        p->plane_bytes[k] = (size_t)p->d.pitch[k] * h * bytes_per_sample;
        if (hipMalloc((void **)&p->d.rec[k], p->plane_bytes[k]) != hipSuccess) {
            svt_amd_set_error("hipMalloc (encode-pass picture) failed");
            return SVT_AMD_ERR_RESOURCES;
        }
    }
    p->d.map_pitch = ((uint32_t)(width >> 2) + 63) & ~63u;
    p->map_bytes = (size_t)p->d.map_pitch * (height >> 2);
    if (hipMalloc((void **)&p->d.mode_map, p->map_bytes) != hipSuccess)
        return SVT_AMD_ERR_RESOURCES;
    p->nlcu = ((width + 63) / 64) * ((height + 63) / 64);
    if (hipMalloc((void **)&p->d_cost, sizeof(SvtAmdCabacCost)) != hipSuccess || hipHostMalloc((void **)&p->h_cost, sizeof(SvtAmdCabacCost), hipHostMallocDefault) != hipSuccess ||
        rate_tables_once(ctx->device))
        return SVT_AMD_ERR_RESOURCES;
    p->d.cost = p->d_cost;
    if (hipMalloc((void **)&p->d_sync, sizeof(unsigned) * (size_t)(1 + 2 * p->nlcu)) != hipSuccess)
        return SVT_AMD_ERR_RESOURCES;
    HIP_TRY(hipMemset(p->d_sync, 0, sizeof(unsigned) * (size_t)(1 + p->nlcu)));
    {   /* ticket -> LCU in wavefront order: anti-diagonals x + 2y (the left and the top-right neighbour lie on the diagonal before),
         * so that a grid no wider than the wavefront keeps every workgroup busy */
        const int wl = (width + 63) / 64, hl = (height + 63) / 64;
        std::vector<unsigned> order;
        order.reserve((size_t)p->nlcu);
        for (int d = 0; d <= (wl - 1) + 2 * (hl - 1); d++)
            for (int y = 0; y < hl; y++) {
                const int x = d - 2 * y;
                if (x >= 0 && x < wl)
                    order.push_back((unsigned)(y * wl + x));
            }
        HIP_TRY(hipMemcpy(p->d_sync + 1 + p->nlcu, order.data(), sizeof(unsigned) * order.size(), hipMemcpyHostToDevice));
    }
    *out = p;
    return SVT_AMD_OK;
}

/* every failure after the object exists releases what was allocated (svt_amd_encdec_picture_destroy tolerates missing members) */
extern "C" int svt_amd_encdec_picture_create(SvtAmdContext *ctx, uint16_t width, uint16_t height, int bytes_per_sample, SvtAmdEncDecPicture **out)
{
    if (out)
        *out = nullptr;
    const int rc = picture_create(ctx, width, height, bytes_per_sample, out);
    if (rc && out && *out) {
        svt_amd_encdec_picture_destroy(ctx, *out);
        *out = nullptr;
    }
    return rc;
}

extern "C" int svt_amd_encdec_picture_begin(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic)
{
    if (!ctx || !pic)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(pic->d.mode_map, 0xFF, pic->map_bytes, ctx->stream)); /* nothing coded yet */
    HIP_TRY(hipStreamSynchronize(ctx->stream));                                   /* other lanes may encode the first LCU */
    pic->deblocked = pic->sao_done = false;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_encdec_picture_destroy(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic)
{
    if (!ctx || !pic)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    (void)hipStreamSynchronize(ctx->stream);
    for (int k = 0; k < 3; k++)
        if (pic->d.rec[k])
            (void)hipFree(pic->d.rec[k]);
    if (pic->d.mode_map)
        (void)hipFree(pic->d.mode_map);
    if (pic->d_sync)
        (void)hipFree(pic->d_sync);
    if (pic->d_order_rect)
        (void)hipFree(pic->d_order_rect);
    if (pic->d_order_md)
        (void)hipFree(pic->d_order_md);
    if (pic->d.prof)
        (void)hipFree(pic->d.prof);
    if (pic->d_cost)
        (void)hipFree(pic->d_cost);
    if (pic->h_cost)
        (void)hipHostFree(pic->h_cost);
    svt_amd_md_state_free(pic);
    if (pic->ev_written)
        (void)hipEventDestroy(pic->ev_written);
    for (int k = 0; k < 3; k++) {
        if (pic->dbk[k])
            (void)hipFree(pic->dbk[k]);
        if (pic->fin[k])
            (void)hipFree(pic->fin[k]);
        if (pic->refp[k])
            (void)hipFree(pic->refp[k]);
    }
    free(pic);
    return SVT_AMD_OK;
}

/* P / B pictures: what the inter units of the picture read besides the LCU records - the reference pictures of list 0 / 1 (device
 * memory, whole padded planes; either may be NULL) and pictureControlSetPtr->cabacCost (HOST pointer, copied on the context's stream).
 * Holds until the next call for this picture object. */
extern "C" int svt_amd_encdec_picture_set_inter(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdRefPicture *ref0,
                                                const SvtAmdRefPicture *ref1, const SvtAmdCabacCost *cost)
{
    if (!ctx || !pic || !cost)
        return SVT_AMD_ERR_BAD_PARAM; /* no reference picture at all: an I picture that needs the rate tables (PM-core quantiser) */
    const SvtAmdRefPicture *refs[2] = {ref0, ref1};
    for (int l = 0; l < 2; l++) {
        const SvtAmdRefPicture *r = refs[l];
        pic->has_ref[l] = r != nullptr;
        memset(&pic->d.ref[l], 0, sizeof(pic->d.ref[l]));
        if (!r)
            continue;
        if (!r->d_y || !r->d_cb || !r->d_cr || r->width != pic->d.width || r->height != pic->d.height || r->originX < 8 || r->originY < 8 ||
            r->strideY < r->width + 2 * r->originX || r->strideC < (r->width + 2 * r->originX) / 2) {
            svt_amd_set_error("svt_amd_encdec_picture_set_inter: reference picture %d does not fit the picture", l);
            return SVT_AMD_ERR_BAD_PARAM;
        }
        EpRefPlanes &E = pic->d.ref[l];
        E.plane[0] = r->d_y, E.plane[1] = r->d_cb, E.plane[2] = r->d_cr;
        E.stride[0] = r->strideY, E.stride[1] = r->strideC;
        E.originX = (int32_t)r->originX, E.originY = (int32_t)r->originY, E.width = (int32_t)r->width, E.height = (int32_t)r->height;
        E.size[0] = (int32_t)(r->strideY * (r->height + 2 * r->originY)), E.size[1] = (int32_t)(r->strideC * ((r->height + 2 * r->originY) / 2));
    }
    HIP_TRY(hipSetDevice(ctx->device));
    memcpy(pic->h_cost, cost, sizeof(*cost)); /* a pageable source would make the copy a staged one the runtime completes inside the call, behind whatever its queue runs */
    HIP_TRY(hipMemcpyAsync(pic->d_cost, pic->h_cost, sizeof(*cost), hipMemcpyHostToDevice, ctx->stream));
    pic->has_cost = true;
    return SVT_AMD_OK;
}

/* hip_encdec_segment: works / results are HOST arrays of n LCUs that do not depend on each other (one wavefront step); blocking.
 * Contexts (lanes) may call concurrently for different LCUs of the same picture as long as the wavefront order holds between calls. */
template <typename WorkT>
static int ep_validate(const SvtAmdEncDecPicture *pic, const WorkT *works, int n, const char *who)
{
    for (int i = 0; i < n; i++) {
        if (works[i].num_cus > SVT_AMD_LCU_MAX_CUS || works[i].lcu_x >= pic->d.width || works[i].lcu_y >= pic->d.height || (works[i].lcu_x & 63) ||
            (works[i].lcu_y & 63)) {
            svt_amd_set_error("%s: bad LCU %d", who, i);
            return SVT_AMD_ERR_BAD_PARAM;
        }
        if (works[i].pm_core && !pic->has_cost) {
            svt_amd_set_error("%s: LCU %d asks for the PM-core quantiser without the picture's rate tables (svt_amd_encdec_picture_set_inter)", who, i);
            return SVT_AMD_ERR_BAD_PARAM;
        }
        for (int c = 0; c < works[i].num_cus; c++) {
            const SvtAmdLcuCu &u = works[i].cu[c];
            const bool inter = u.pred_mode == 1;
            if ((u.pred_mode != 2 && !inter) || !(u.size == 8 || u.size == 16 || u.size == 32 || (inter && u.size == 64)) || u.intra_luma_mode > 34 ||
                (u.x & (u.size - 1)) || (u.y & (u.size - 1)) || u.x + u.size > 64 || u.y + u.size > 64 || works[i].lcu_x + u.x + u.size > pic->d.width ||
                works[i].lcu_y + u.y + u.size > pic->d.height) {
                svt_amd_set_error("%s: LCU %d unit %d is not an intra 2Nx2N unit of 8..32 or an inter 2Nx2N unit of 8..64 inside the picture", who, i, c);
                return SVT_AMD_ERR_BAD_PARAM;
            }
            if (inter && (u.inter_dir > 2 || u.inter_kind > SVT_AMD_EP_INTER_SKIP || (u.inter_dir != 1 && !pic->has_ref[0]) || (u.inter_dir != 0 && !pic->has_ref[1]))) {
                svt_amd_set_error("%s: LCU %d unit %d: inter unit without its reference picture (svt_amd_encdec_picture_set_inter) or with a bad direction / kind",
                                  who, i, c);
                return SVT_AMD_ERR_BAD_PARAM;
            }
        }
    }
    return SVT_AMD_OK;
}

template <typename T>
static int encode_lcus(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const typename EpTypes<T>::Work *works, int n, typename EpTypes<T>::Result *results,
                       const char *who)
{
    typedef typename EpTypes<T>::Work WorkT;
    typedef typename EpTypes<T>::Result ResultT;
    if (!ctx || !pic || !works || !results || n < 1 || n > 1024)
        return SVT_AMD_ERR_BAD_PARAM;
    if (pic->d.bps != sizeof(T)) {
        svt_amd_set_error("%s: the picture holds %u-byte samples", who, pic->d.bps);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    int rv = ep_validate(pic, works, n, who);
    if (rv)
        return rv;
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *d = nullptr;
    const size_t wb = sizeof(WorkT) * (size_t)n, rb = sizeof(ResultT) * (size_t)n, wba = (wb + 255) & ~(size_t)255;
    int rc = svt_amd_ctx_scratch(ctx, wba + rb, &d);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(d, works, wb, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_encode_lcu<T>, dim3((unsigned)n), dim3(256), 0, ctx->stream, pic->d, (const WorkT *)d, (ResultT *)(d + wba));
    HIP_TRY(hipGetLastError());
    if ((rc = ep_picture_written(ctx, pic)) != 0)
        return rc;
    HIP_TRY(hipMemcpyAsync(results, d + wba, rb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_encode_lcus(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works, int n, SvtAmdLcuResult *results)
{
    return encode_lcus<uint8_t>(ctx, pic, works, n, results, "svt_amd_encode_lcus");
}
extern "C" int svt_amd_encode_lcus16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works, int n, SvtAmdLcuResult16 *results)
{
    return encode_lcus<uint16_t>(ctx, pic, works, n, results, "svt_amd_encode_lcus16");
}

/* One call per picture: works / results are HOST arrays of ALL LCUs of the picture in raster order; the wavefront runs on the
 * device (k_encode_picture).  d_works / d_results (optional, device) replace the host arrays: nothing crosses PCIe then. */
template <typename T>
static int encode_picture(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const typename EpTypes<T>::Work *works, typename EpTypes<T>::Result *results,
                          const typename EpTypes<T>::Work *d_works, typename EpTypes<T>::Result *d_results, int parallel_tiles, int free_lcus = 0,
                          const SvtAmdRect *rect = nullptr)
{
    typedef typename EpTypes<T>::Work WorkT;
    typedef typename EpTypes<T>::Result ResultT;
    if (pic->d.bps != sizeof(T)) {
        svt_amd_set_error("svt_amd_encode_picture: the picture holds %u-byte samples", pic->d.bps);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    const int n = pic->nlcu, wl = (pic->d.width + 63) / 64;
    HIP_TRY(hipSetDevice(ctx->device));
    int n_active = n;
    const unsigned *d_order = pic->d_sync + 1 + n;
    if (rect) {
        /* a rank's share (multi-GPU): only the LCUs of a rectangle of whole tiles are drawn, in the same anti-diagonal order */
        const int x0 = rect->x / 64, y0 = rect->y / 64, x1 = (rect->x + rect->w + 63) / 64, y1 = (rect->y + rect->h + 63) / 64, hl0 = (pic->d.height + 63) / 64;
        if ((rect->x & 63) || (rect->y & 63) || !rect->w || !rect->h || x1 > wl || y1 > hl0 || !works) {
            svt_amd_set_error("svt_amd_encode_picture_rect: the rectangle is not a set of whole LCUs of the picture");
            return SVT_AMD_ERR_BAD_PARAM;
        }
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                const auto &W = works[y * wl + x];
                const bool below_own = y + 1 < y1, below_any = y + 1 < hl0;
                if ((x == x0 && !W.tile_left) || (y == y0 && !W.tile_top) || (x == x1 - 1 && x1 < wl && !W.tile_right) ||
                    (!below_own && below_any && !works[(y + 1) * wl + x].tile_top)) {
                    svt_amd_set_error("svt_amd_encode_picture_rect: the rectangle's border at LCU (%d, %d) is not a tile border", x, y);
                    return SVT_AMD_ERR_BAD_PARAM;
                }
            }
        std::vector<unsigned> order;
        for (int d = 0; d <= (x1 - x0 - 1) + 2 * (y1 - y0 - 1); d++)
            for (int y = y0; y < y1; y++) {
                const int x = x0 + d - 2 * (y - y0);
                if (x >= x0 && x < x1)
                    order.push_back((unsigned)(y * wl + x));
            }
        n_active = (int)order.size();
        if (!pic->d_order_rect && hipMalloc((void **)&pic->d_order_rect, sizeof(unsigned) * (size_t)n) != hipSuccess)
            return SVT_AMD_ERR_RESOURCES;
        HIP_TRY(hipMemcpyAsync(pic->d_order_rect, order.data(), sizeof(unsigned) * order.size(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream)); /* the list is a local */
        d_order = pic->d_order_rect;
    }
    if (!d_works) {
        for (int i = 0; i < n; i++)
            if (works[i].lcu_x != (i % wl) * 64 || works[i].lcu_y != (i / wl) * 64) {
                svt_amd_set_error("svt_amd_encode_picture: LCU %d is not at raster position %d", i, i);
                return SVT_AMD_ERR_BAD_PARAM;
            }
        int rv = ep_validate(pic, works, n, "svt_amd_encode_picture");
        if (rv)
            return rv;
        parallel_tiles = 0; /* tiles = LCUs that wait for nobody (top-left corners) */
        for (int i = 0; i < n; i++) {
            parallel_tiles += works[i].tile_left && works[i].tile_top;
            bool intra = false;
            for (int k = 0; k < works[i].num_cus; k++)
                intra |= works[i].cu[k].pred_mode == 2;
            free_lcus += !intra;
        }
        uint8_t *d = nullptr;
        const size_t wb = sizeof(WorkT) * (size_t)n, rb = sizeof(ResultT) * (size_t)n, wba = (wb + 255) & ~(size_t)255;
        int rc = svt_amd_ctx_scratch(ctx, wba + rb, &d);
        if (rc)
            return rc;
        HIP_TRY(hipMemcpyAsync(d, works, wb, hipMemcpyHostToDevice, ctx->stream));
        d_works = (const WorkT *)d, d_results = (ResultT *)(d + wba);
        if (rect) /* the other ranks' LCUs: nothing coded here (the filters behind the encode pass read every LCU's flags) */
            HIP_TRY(hipMemsetAsync(d + wba, 0, rb, ctx->stream));
    }
    pic->epoch++;
    pic->deblocked = pic->sao_done = false;
    HIP_TRY(hipMemsetAsync(pic->d_sync, 0, sizeof(unsigned), ctx->stream));              /* ticket counter */
    HIP_TRY(hipMemsetAsync(pic->d.mode_map, 0xFF, pic->map_bytes, ctx->stream));          /* nothing coded yet */
    /* persistent grid = the widest wavefront (an LCU row advances two LCUs behind the row above) of every tile that can run on its
     * own: more workgroups would only poll, and they would hold the CUs other pictures' launches could use */
    const int hl = (pic->d.height + 63) / 64;
    int grid = ((wl + 1) / 2 < hl ? (wl + 1) / 2 : hl) * (parallel_tiles > 0 ? parallel_tiles : 1) + 1;
    if (free_lcus * 2 > n) /* a P / B picture: most LCUs have no intra unit and wait for nobody - as many workgroups as the device holds */
        grid = 512;
    grid = grid > n_active ? n_active : grid > 512 ? 512 : grid;
    hipLaunchKernelGGL(k_encode_picture<T>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, pic->d, d_works, d_results, n_active, wl, pic->d_sync, pic->d_sync + 1, d_order, pic->epoch);
    HIP_TRY(hipGetLastError());
    {
        const int rcw = ep_picture_written(ctx, pic);
        if (rcw)
            return rcw;
    }
    if (results)
        HIP_TRY(hipMemcpyAsync(results, d_results, sizeof(ResultT) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    return SVT_AMD_OK;
}

extern "C" int svt_amd_encode_picture(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works, SvtAmdLcuResult *results)
{
    if (!ctx || !pic || !works || !results)
        return SVT_AMD_ERR_BAD_PARAM;
    int rc = encode_picture<uint8_t>(ctx, pic, works, results, nullptr, nullptr, 0);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_encode_picture16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works, SvtAmdLcuResult16 *results)
{
    if (!ctx || !pic || !works || !results)
        return SVT_AMD_ERR_BAD_PARAM;
    int rc = encode_picture<uint16_t>(ctx, pic, works, results, nullptr, nullptr, 0);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}

/* A rank's share of the picture (multi-GPU over tiles; Codec/EbEncDecProcess.c:2743-2760: EncDec never reads across a tile edge): works /
 * results as in svt_amd_encode_picture, only the LCUs inside `rect` (a rectangle of whole tiles, svt_amd_tile_partition) are encoded; the
 * other LCUs' results come back zeroed and their part of the device picture is whatever it was. */
extern "C" int svt_amd_encode_picture_rect(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works, SvtAmdLcuResult *results,
                                           const SvtAmdRect *rect)
{
    if (!ctx || !pic || !works || !results || !rect)
        return SVT_AMD_ERR_BAD_PARAM;
    int rc = encode_picture<uint8_t>(ctx, pic, works, results, nullptr, nullptr, 0, 0, rect);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_encode_picture_rect16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works, SvtAmdLcuResult16 *results,
                                             const SvtAmdRect *rect)
{
    if (!ctx || !pic || !works || !results || !rect)
        return SVT_AMD_ERR_BAD_PARAM;
    int rc = encode_picture<uint16_t>(ctx, pic, works, results, nullptr, nullptr, 0, 0, rect);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}

/* The rectangle (whole tiles, svt_amd_tile_partition) the object's MODE-DECISION calls work on: svt_amd_md_encode_picture[_inter] then draws the tickets of its
 * LCUs only (the same anti-diagonal order), decides and encodes them, and leaves the other LCUs' records zeroed.  NULL = the whole picture again. */
extern "C" int svt_amd_encdec_picture_set_rect(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdRect *rect)
{
    if (!ctx || !pic)
        return SVT_AMD_ERR_BAD_PARAM;
    if (!rect) {
        pic->md_rect_n = 0;
        return SVT_AMD_OK;
    }
    const int wl = (pic->d.width + 63) / 64, hl = (pic->d.height + 63) / 64;
    const int x0 = rect->x / 64, y0 = rect->y / 64, x1 = (rect->x + rect->w + 63) / 64, y1 = (rect->y + rect->h + 63) / 64;
    if ((rect->x & 63) || (rect->y & 63) || !rect->w || !rect->h || x1 > wl || y1 > hl) {
        svt_amd_set_error("svt_amd_encdec_picture_set_rect: the rectangle is not a set of whole LCUs of the picture");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<unsigned> order;
    for (int d = 0; d <= (x1 - x0 - 1) + 2 * (y1 - y0 - 1); d++)
        for (int y = y0; y < y1; y++) {
            const int x = x0 + d - 2 * (y - y0);
            if (x >= x0 && x < x1)
                order.push_back((unsigned)(y * wl + x));
        }
    if (!pic->d_order_md && hipMalloc((void **)&pic->d_order_md, sizeof(unsigned) * (size_t)pic->nlcu) != hipSuccess)
        return SVT_AMD_ERR_RESOURCES;
    HIP_TRY(hipMemcpy(pic->d_order_md, order.data(), sizeof(unsigned) * order.size(), hipMemcpyHostToDevice));
    pic->md_rect_n = (int)order.size(), pic->md_rect = *rect;
    return SVT_AMD_OK;
}

/* The picture object's latest stage (after SAO, else deblocked, else as encoded) completed across ranks: own rectangle out, everybody else's in
 * (svt_amd_recon_exchange on the object's planes; call it before svt_amd_encdec_picture_reference pads the picture).  The _pack form is the local
 * half for hosts with their own transport (svt_amd_recon_pack). */
extern "C" int svt_amd_encdec_picture_exchange(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdRect *rects, int world, int rank)
{
    if (!ctx || !pic || !rects)
        return SVT_AMD_ERR_BAD_PARAM;
    uint8_t *const *stage = pic->sao_done ? pic->fin : pic->deblocked ? pic->dbk : pic->d.rec;
    void *planes[3] = {stage[0], stage[1], stage[2]};
    const uint32_t pitch[3] = {pic->d.pitch[0] * pic->d.bps, pic->d.pitch[1] * pic->d.bps, pic->d.pitch[2] * pic->d.bps};
    const int rc = svt_amd_recon_exchange(ctx, planes, pitch, (int)pic->d.bps, rects, world, rank);
    /* "an exchange that fills it": a reader on another context's stream (svt_amd_encdec_picture_import) orders itself behind the all-gather, not behind whatever wrote the
     * object before it */
    return rc ? rc : ep_picture_written(ctx, pic);
}
/* Picture-level parallelism: the rank that owns a picture encodes all of it; its finished picture (latest stage) goes to every rank ONCE, before
 * svt_amd_encdec_picture_reference pads it there.  On the receiving ranks the planes land in the object's final stage (the object then counts as deblocked and
 * SAO-filtered: nothing else of the picture exists on them).  _broadcast: ncclBroadcast on the context's communicator; _import: the same hand-over between two
 * picture objects of one process (logical ranks on one device, peers with direct access, or a host with its own transport), a device-to-device copy. */
extern "C" int svt_amd_recon_broadcast(SvtAmdContext *ctx, void *const d_planes[3], const size_t bytes[3], int world, int rank, int root);
static int final_stage_planes(SvtAmdEncDecPicture *pic) /* the final stage's planes exist from the first SAO call, or from here */
{
    for (int k = 0; k < 3; k++)
        if (!pic->fin[k] && hipMalloc((void **)&pic->fin[k], pic->plane_bytes[k]) != hipSuccess) {
            svt_amd_set_error("hipMalloc (finished picture) failed");
            return SVT_AMD_ERR_RESOURCES;
        }
    return SVT_AMD_OK;
}
extern "C" int svt_amd_encdec_picture_broadcast(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, int world, int rank, int root)
{
    if (!ctx || !pic || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world)
        return SVT_AMD_ERR_BAD_PARAM;
    if (world == 1)
        return SVT_AMD_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rank != root ? final_stage_planes(pic) : SVT_AMD_OK;
    if (rc)
        return rc;
    uint8_t *const *stage = rank != root ? pic->fin : pic->sao_done ? pic->fin : pic->deblocked ? pic->dbk : pic->d.rec;
    void *planes[3] = {stage[0], stage[1], stage[2]};
    rc = svt_amd_recon_broadcast(ctx, planes, pic->plane_bytes, world, rank, root);
    if (rc == SVT_AMD_OK && rank != root) { /* the receiver's object now holds the root's finished picture: nothing of its own earlier picture is valid any more */
        pic->deblocked = pic->sao_done = true;
        pic->epoch++;
        rc = ep_picture_written(ctx, pic);
    }
    return rc;
}
extern "C" int svt_amd_encdec_picture_import(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdEncDecPicture *from)
{
    if (!ctx || !pic || !from || pic == from || pic->d.width != from->d.width || pic->d.height != from->d.height || pic->d.bps != from->d.bps)
        return SVT_AMD_ERR_BAD_PARAM;
    if (!from->written || !from->ev_written) {
        svt_amd_set_error("svt_amd_encdec_picture_import: the source object holds no encoded picture");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = final_stage_planes(pic);
    if (rc)
        return rc;
    /* behind whatever the owner's stream still has queued on the source picture (its encode pass / filters run on another context, possibly another device) */
    HIP_TRY(hipStreamWaitEvent(ctx->stream, from->ev_written, 0));
    uint8_t *const *stage = from->sao_done ? from->fin : from->deblocked ? from->dbk : from->d.rec;
    for (int p = 0; p < 3; p++) /* hipMemcpyDefault: the source may live on a peer device */
        HIP_TRY(hipMemcpyAsync(pic->fin[p], stage[p], pic->plane_bytes[p], hipMemcpyDefault, ctx->stream));
    pic->deblocked = pic->sao_done = true;
    pic->epoch++; /* the per-LCU completion marks of the receiver's own earlier picture do not describe this one */
    return ep_picture_written(ctx, pic);
}
extern "C" int svt_amd_encdec_picture_pack(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdRect *rects, int world, int r, void *d_slots,
                                           size_t slot_bytes, int to_slot)
{
    if (!ctx || !pic || !rects)
        return SVT_AMD_ERR_BAD_PARAM;
    uint8_t *const *stage = pic->sao_done ? pic->fin : pic->deblocked ? pic->dbk : pic->d.rec;
    void *planes[3] = {stage[0], stage[1], stage[2]};
    const uint32_t pitch[3] = {pic->d.pitch[0] * pic->d.bps, pic->d.pitch[1] * pic->d.bps, pic->d.pitch[2] * pic->d.bps};
    return svt_amd_recon_pack(ctx, planes, pitch, (int)pic->d.bps, rects, world, r, d_slots, slot_bytes, to_slot);
}

/* Device-resident form: work and result arrays already in HBM (what a device-side mode decision would leave there); asynchronous on
 * the context's stream.  The caller vouches for the unit lists (no host copy to validate). */
extern "C" int svt_amd_encode_picture_device(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *d_works, SvtAmdLcuResult *d_results,
                                             int tiles)
{
    if (!ctx || !pic || !d_works || !d_results || tiles < 1)
        return SVT_AMD_ERR_BAD_PARAM;
    return encode_picture<uint8_t>(ctx, pic, nullptr, nullptr, d_works, d_results, tiles);
}
/* the same with the caller's knowledge of the picture: free_lcus = LCUs without an intra unit (they wait for no neighbour; sizes the grid) */
extern "C" int svt_amd_encode_picture_device_inter(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *d_works, SvtAmdLcuResult *d_results,
                                                   int tiles, int free_lcus)
{
    if (!ctx || !pic || !d_works || !d_results || tiles < 1 || free_lcus < 0)
        return SVT_AMD_ERR_BAD_PARAM;
    return encode_picture<uint8_t>(ctx, pic, nullptr, nullptr, d_works, d_results, tiles, free_lcus);
}

/* ---- deblocking behind the encode pass ------------------------------------------------------------------------------------------ */
/* Once every LCU of the picture is encoded (svt_amd_encode_picture, or LCU by LCU), the device picture goes through the
 * picture-level boundary-strength and deblocking kernels (filter_kernels.hip, proven on recorded pictures) IN PLACE: what the
 * reference's reconstruction holds after its per-LCU drivers - the finished picture when SAO is off.  The two maps those kernels
 * read (coding unit per 8x8 block, luma cbf per 4x4 block) and the QP array come from the contract records the host already has. */
template <typename T>
static int picture_deblock(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const typename EpTypes<T>::Work *works, const typename EpTypes<T>::Result *results,
                           const SvtAmdDeblockParams *prm, void *out_y, void *out_cb, void *out_cr)
{
    if (!ctx || !pic || !works || !results || !prm || pic->d.bps != sizeof(T))
        return SVT_AMD_ERR_BAD_PARAM;
    const uint32_t w = pic->d.width, h = pic->d.height, wl = (w + 63) / 64, nlcu = (uint32_t)pic->nlcu;
    const uint32_t w8 = w / 8, h8 = h / 8, w4 = w / 4, h4 = h / 4;
    std::vector<SvtAmdCuMapEntry> map((size_t)w8 * h8);
    std::vector<uint8_t> cbf((size_t)w4 * h4), qp((size_t)w8 * h8), edge(nlcu);
    ::memset(map.data(), 0, map.size() * sizeof(SvtAmdCuMapEntry));
    for (uint32_t i = 0; i < nlcu; i++) {
        const auto &W = works[i];
        if (W.lcu_x != (i % wl) * 64 || W.lcu_y != (i / wl) * 64) {
            svt_amd_set_error("svt_amd_encdec_picture_deblock: LCU %u is not at raster position %u", i, i);
            return SVT_AMD_ERR_BAD_PARAM;
        }
        edge[i] = (uint8_t)((W.tile_left ? 1 : 0) | (W.tile_top ? 2 : 0));
        for (int c = 0; c < W.num_cus; c++) {
            const SvtAmdLcuCu &u = W.cu[c];
            const uint32_t x0 = W.lcu_x + u.x, y0 = W.lcu_y + u.y;
            if (x0 + u.size > w || y0 + u.size > h || u.size < 8)
                return SVT_AMD_ERR_BAD_PARAM;
            SvtAmdCuMapEntry e;
            ::memset(&e, 0, sizeof(e));
            e.mode = u.pred_mode, e.size_log2 = (uint8_t)(u.size == 8 ? 3 : u.size == 16 ? 4 : u.size == 32 ? 5 : 6);
            if (u.pred_mode == 1) /* inter: the prediction unit's direction and motion vectors decide the strength of its edges */
                e.dir = u.inter_dir, ::memcpy(e.mv, u.mv, sizeof(e.mv));
            for (uint32_t y = y0 / 8; y < (y0 + u.size) / 8; y++)
                for (uint32_t x = x0 / 8; x < (x0 + u.size) / 8; x++)
                    map[(size_t)y * w8 + x] = e, qp[(size_t)y * w8 + x] = u.qp;
            if (u.size == 64) { /* four 32x32 transform units: result entries c + 1 .. c + 4 */
                for (uint32_t y = 0; y < 16; y++)
                    for (uint32_t t = 0; t < 2; t++)
                        ::memset(&cbf[(size_t)(y0 / 4 + y) * w4 + x0 / 4 + 8 * t], results[i].cu[c + 1 + 2 * (y >> 3) + t].cbf[0], 8);
            } else {
                for (uint32_t y = y0 / 4; y < (y0 + u.size) / 4; y++)
                    ::memset(&cbf[(size_t)y * w4 + x0 / 4], results[i].cu[c].cbf[0], u.size / 4);
            }
        }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t b_map = (map.size() * sizeof(SvtAmdCuMapEntry) + 255) & ~(size_t)255, b_cbf = (cbf.size() + 255) & ~(size_t)255,
                 b_qp = (qp.size() + 255) & ~(size_t)255, b_edge = ((size_t)nlcu + 255) & ~(size_t)255, b_bs = (size_t)nlcu * 256;
    uint8_t *d = nullptr;
    int rc = svt_amd_ctx_scratch(ctx, b_map + b_cbf + b_qp + b_edge + 2 * b_bs, &d);
    if (rc)
        return rc;
    uint8_t *d_map = d, *d_cbf = d_map + b_map, *d_qp = d_cbf + b_cbf, *d_edge = d_qp + b_qp, *d_bsv = d_edge + b_edge, *d_bsh = d_bsv + b_bs;
    HIP_TRY(hipMemcpyAsync(d_map, map.data(), map.size() * sizeof(SvtAmdCuMapEntry), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_cbf, cbf.data(), cbf.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_qp, qp.data(), qp.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_edge, edge.data(), edge.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream)); /* the host vectors back the copies */
    if ((rc = svt_amd_bs_picture(ctx, (const SvtAmdCuMapEntry *)d_map, d_cbf, w, h, prm->slice_type, prm->ref_poc[0], prm->ref_poc[1], d_edge, d_bsv,
                                 d_bsh)) != 0)
        return rc;
    if (pic->d.pitch[1] != pic->d.pitch[2])
        return SVT_AMD_ERR_BAD_PARAM;
    /* the deblocked picture is a second set of planes: the un-deblocked one stays (the neighbours of LCUs still to come, and the part of
     * the SAO statistics that the reference gathers before an LCU's right / bottom edges are filtered) */
    for (int k = 0; k < 3; k++) {
        if (!pic->dbk[k] && hipMalloc((void **)&pic->dbk[k], pic->plane_bytes[k]) != hipSuccess) {
            svt_amd_set_error("hipMalloc (deblocked picture) failed");
            return SVT_AMD_ERR_RESOURCES;
        }
        HIP_TRY(hipMemcpyAsync(pic->dbk[k], pic->d.rec[k], pic->plane_bytes[k], hipMemcpyDeviceToDevice, ctx->stream));
    }
    if ((rc = svt_amd_dlf_picture(ctx, (int)sizeof(T), pic->dbk[0], pic->d.pitch[0], pic->dbk[1], pic->dbk[2], pic->d.pitch[1], w, h, d_bsv, d_bsh,
                                  d_qp, w8, prm->tc_offset, prm->beta_offset, prm->cb_qp_offset, prm->cr_qp_offset)) != 0)
        return rc;
    pic->deblocked = true;
    if ((rc = ep_picture_written(ctx, pic)) != 0)
        return rc;
    void *outs[3] = {out_y, out_cb, out_cr};
    for (int k = 0; k < 3; k++)
        if (outs[k]) {
            const uint32_t pw = k ? w / 2 : w, ph = k ? h / 2 : h;
            HIP_TRY(hipMemcpy2DAsync(outs[k], (size_t)pw * sizeof(T), pic->dbk[k], (size_t)pic->d.pitch[k] * sizeof(T), (size_t)pw * sizeof(T), ph,
                                     hipMemcpyDeviceToHost, ctx->stream));
        }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}

/* ---- SAO behind the deblocked device picture ------------------------------------------------------------------------------------ */
/* the LCUs' source samples as planes (the statistics kernels read planes) */
template <typename T>
__global__ __launch_bounds__(256) void k_ep_source_planes(const typename EpTypes<T>::Work *__restrict__ works, T *sy, T *scb, T *scr, int pitchY, int pitchC,
                                                          int width, int height)
{
    const typename EpTypes<T>::Work &W = works[blockIdx.x];
    const int lw = min(64, width - (int)W.lcu_x), lh = min(64, height - (int)W.lcu_y);
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int y = i >> 6, x = i & 63;
        if (x < lw && y < lh)
            sy[(size_t)(W.lcu_y + y) * pitchY + W.lcu_x + x] = W.src_y[i];
    }
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int y = i >> 5, x = i & 31;
        if (x < lw / 2 && y < lh / 2) {
            scb[(size_t)(W.lcu_y / 2 + y) * pitchC + W.lcu_x / 2 + x] = W.src_cb[i];
            scr[(size_t)(W.lcu_y / 2 + y) * pitchC + W.lcu_x / 2 + x] = W.src_cr[i];
        }
    }
}

/* The picture as the reference's SaoGenerationDecision sees each LCU (EbCodingLoop.c:4600-4750): the LCU's own deblocking drivers have run,
 * those of the LCUs to its right and below have not.  Those later drivers own the 8x8 filter blocks centred on the LCU boundary
 * (LCUBoundaryDLFCore, EbDeblockingFilter.c:2828), i.e. every edge segment inside the last 4 columns / rows of the LCU (in the plane's own
 * samples) and nothing else inside it: the LCU is the deblocked picture with those strips still un-deblocked, where a neighbour OF THE SAME
 * TILE follows (at picture and tile edges the LCU's own LCUPictureEdgeDLFCore has finished them; follow: bit 0 right, bit 1 below).
 * (tests/test_oracle_encodepass_golden.py::test_encoder_order_sao_statistics_from_two_pictures proves it on the encoder's own records.) */
template <typename T>
__global__ __launch_bounds__(256) void k_ep_sao_composite(const T *__restrict__ dbk, const T *__restrict__ rec, T *__restrict__ out, int pitch, int w, int h,
                                                          int lg_lcu, int wl, const uint8_t *__restrict__ follow)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, lcu = 1 << lg_lcu;
    if (x >= w)
        return;
    const int f = follow[(y >> lg_lcu) * wl + (x >> lg_lcu)];
    const bool right = (x & (lcu - 1)) >= lcu - 4 && (f & 1), bottom = (y & (lcu - 1)) >= lcu - 4 && (f & 2);
    const size_t o = (size_t)y * pitch + x;
    out[o] = (right || bottom) ? rec[o] : dbk[o];
}

/* Statistics of every LCU in the encoder's order, the parameter decision of the whole picture (merge wavefront) and the application, behind
 * svt_amd_encdec_picture_deblock: what is left in the picture object (and copied out) is the encoder's finished reconstruction. */
template <typename T>
static int picture_sao(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const typename EpTypes<T>::Work *works, const SvtAmdSaoDecisionParams *prm,
                       const uint8_t *enable, SvtAmdSaoLcuParams *lcu_out, void *out_y, void *out_cb, void *out_cr, bool apply)
{
    typedef typename EpTypes<T>::Work WorkT;
    if (!ctx || !pic || !works || !prm || pic->d.bps != sizeof(T))
        return SVT_AMD_ERR_BAD_PARAM;
    if (!pic->deblocked && apply) {
        svt_amd_set_error("svt_amd_encdec_picture_sao: svt_amd_encdec_picture_deblock comes first");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    const uint32_t w = pic->d.width, h = pic->d.height, wl = (w + 63) / 64, hl = (h + 63) / 64, nlcu = (uint32_t)pic->nlcu;
    std::vector<SvtAmdSaoLcuParams> lp(nlcu);
    std::vector<uint8_t> follow(nlcu);
    ::memset(lp.data(), 0, nlcu * sizeof(SvtAmdSaoLcuParams));
    for (uint32_t i = 0; i < nlcu; i++) {
        if (works[i].lcu_x != (i % wl) * 64 || works[i].lcu_y != (i / wl) * 64)
            return SVT_AMD_ERR_BAD_PARAM;
        const bool bottom_edge = i + wl >= nlcu || works[i + wl].tile_top;
        lp[i].edge_flags = (uint8_t)((works[i].tile_left ? 1 : 0) | (works[i].tile_right ? 2 : 0) | (works[i].tile_top ? 4 : 0) | (bottom_edge ? 8 : 0));
        follow[i] = (uint8_t)(((i % wl) + 1 < wl && !works[i].tile_right ? 1 : 0) | (!bottom_edge ? 2 : 0));
    }
    HIP_TRY(hipSetDevice(ctx->device));
    for (int k = 0; k < 3 && apply; k++)
        if (!pic->fin[k] && hipMalloc((void **)&pic->fin[k], pic->plane_bytes[k]) != hipSuccess)
            return SVT_AMD_ERR_RESOURCES;
    auto up = [](size_t n) { return (n + 255) & ~(size_t)255; };
    const size_t b_works = up(sizeof(WorkT) * nlcu), b_pl[3] = {up(pic->plane_bytes[0]), up(pic->plane_bytes[1]), up(pic->plane_bytes[2])};
    const size_t b_stats = up(sizeof(SvtAmdSaoStats) * nlcu), b_par = up(sizeof(SvtAmdSaoLcuParams) * nlcu), b_cost = up(16 * (size_t)nlcu), b_en = up(nlcu);
    uint8_t *d = nullptr;
    int rc = svt_amd_ctx_scratch(ctx, b_works + 2 * (b_pl[0] + b_pl[1] + b_pl[2]) + 3 * b_stats + b_par + b_cost + 2 * b_en, &d);
    if (rc)
        return rc;
    uint8_t *d_works = d, *d_src[3], *d_cmp[3], *q = d + b_works;
    for (int k = 0; k < 3; k++)
        d_src[k] = q, q += b_pl[k];
    for (int k = 0; k < 3; k++)
        d_cmp[k] = q, q += b_pl[k];
    SvtAmdSaoStats *d_stats[3];
    for (int k = 0; k < 3; k++)
        d_stats[k] = (SvtAmdSaoStats *)q, q += b_stats;
    SvtAmdSaoLcuParams *d_par = (SvtAmdSaoLcuParams *)q;
    q += b_par;
    int64_t *d_cost = (int64_t *)q;
    q += b_cost;
    uint8_t *d_en = q, *d_follow = q + b_en;
    HIP_TRY(hipMemcpyAsync(d_works, works, sizeof(WorkT) * nlcu, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_follow, follow.data(), nlcu, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_par, lp.data(), sizeof(SvtAmdSaoLcuParams) * nlcu, hipMemcpyHostToDevice, ctx->stream));
    if (enable)
        HIP_TRY(hipMemcpyAsync(d_en, enable, nlcu, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_stats[0], 0, 3 * b_stats, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream)); /* lp backs its copy */
    const int pY = (int)pic->d.pitch[0], pC = (int)pic->d.pitch[1];
    hipLaunchKernelGGL(k_ep_source_planes<T>, dim3(nlcu), dim3(256), 0, ctx->stream, (const WorkT *)d_works, (T *)d_src[0], (T *)d_src[1], (T *)d_src[2], pY, pC,
                       (int)w, (int)h);
    for (int k = 0; k < 3 && pic->deblocked; k++) {
        const int pw = k ? w / 2 : w, ph = k ? h / 2 : h;
        hipLaunchKernelGGL(k_ep_sao_composite<T>, dim3((pw + 255) / 256, ph), dim3(256), 0, ctx->stream, (const T *)pic->dbk[k], (const T *)pic->d.rec[k],
                           (T *)d_cmp[k], k ? pC : pY, pw, ph, k ? 5 : 6, (int)wl, (const uint8_t *)d_follow);
    }
    HIP_TRY(hipGetLastError());
    /* GatherSaoStatisticsLcu* of the components the mode looks at (EbSampleAdaptiveOffsetGenerationDecision.c:647-760) */
    const int ncomp = prm->mm_sao ? 3 : (prm->temporal_layer < 2 ? 1 : 0);
    for (int k = 0; k < ncomp; k++)
        if ((rc = svt_amd_sao_gather_picture(ctx, (int)sizeof(T), d_src[k], k ? pC : pY, pic->deblocked ? (const void *)d_cmp[k] : (const void *)pic->d.rec[k],
                                             k ? pC : pY, k ? w / 2 : w, k ? h / 2 : h, k ? 32 : 64, prm->mm_sao ? 0 : 1, d_stats[k])) != 0)
            return rc; /* a picture that is never deblocked (decide-only call) is its own encoder-order view */
    if ((rc = svt_amd_sao_decide_picture(ctx, prm, d_stats[0], d_stats[1], d_stats[2], wl, hl, enable ? d_en : nullptr, d_par, d_cost)) != 0)
        return rc;
    const void *srcs[3] = {pic->dbk[0], pic->dbk[1], pic->dbk[2]};
    void *dsts[3] = {pic->fin[0], pic->fin[1], pic->fin[2]};
    if (apply && (rc = svt_amd_sao_apply_picture(ctx, (int)sizeof(T), srcs, dsts, pic->d.pitch[0], pic->d.pitch[1], w, h, d_par, 1, 1)) != 0)
        return rc;
    if (lcu_out)
        HIP_TRY(hipMemcpyAsync(lcu_out, d_par, sizeof(SvtAmdSaoLcuParams) * nlcu, hipMemcpyDeviceToHost, ctx->stream));
    void *outs[3] = {out_y, out_cb, out_cr};
    for (int k = 0; k < 3 && apply; k++)
        if (outs[k]) {
            const uint32_t pw = k ? w / 2 : w, ph = k ? h / 2 : h;
            HIP_TRY(hipMemcpy2DAsync(outs[k], (size_t)pw * sizeof(T), pic->fin[k], (size_t)pic->d.pitch[k] * sizeof(T), (size_t)pw * sizeof(T), ph,
                                     hipMemcpyDeviceToHost, ctx->stream));
        }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    pic->sao_done = pic->sao_done || apply;
    return ep_picture_written(ctx, pic);
}

/* PadRefAndSetFlags (Codec/EbEncDecProcess.c:1805; GeneratePadding / GeneratePadding16Bit): the finished picture inside a frame of
 * replicated edge samples */
template <typename T>
__global__ __launch_bounds__(256) void k_ep_pad(const T *__restrict__ src, int spitch, int w, int h, T *__restrict__ dst, int dstride, int ox, int oy)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w + 2 * ox)
        return;
    dst[(size_t)y * dstride + x] = src[(size_t)min(max(y - oy, 0), h - 1) * spitch + min(max(x - ox, 0), w - 1)];
}

/* The picture object's latest stage (after SAO, else deblocked, else as encoded) as a padded reference picture in HBM: what
 * svt_amd_encdec_picture_set_inter of a later picture takes - reference pictures never leave the device.  origin_x / origin_y: luma padding
 * (the reference uses LCU size + 16 = 80); the planes are (width + 2 origin_x) samples wide, chroma half of everything.  out_*: optional
 * HOST copies of the padded planes.  The planes live until the picture object is destroyed or the call is repeated. */
extern "C" int svt_amd_encdec_picture_reference(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, uint32_t origin_x, uint32_t origin_y, SvtAmdRefPicture *ref,
                                                void *out_y, void *out_cb, void *out_cr)
{
    if (!ctx || !pic || !ref || origin_x < 8 || origin_y < 8 || (origin_x & 1) || (origin_y & 1) || origin_x > 256 || origin_y > 256)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t w = pic->d.width, h = pic->d.height, bps = pic->d.bps;
    uint8_t *const *stage = pic->sao_done ? pic->fin : pic->deblocked ? pic->dbk : pic->d.rec;
    void *outs[3] = {out_y, out_cb, out_cr};
    for (int k = 0; k < 3; k++) {
        const int sh = k ? 1 : 0, pw = (int)(w >> sh), ph = (int)(h >> sh), ox = (int)(origin_x >> sh), oy = (int)(origin_y >> sh), stride = pw + 2 * ox, rows = ph + 2 * oy;
        const size_t need = (size_t)stride * rows * bps;
        if (pic->refp_bytes[k] < need) {
            if (pic->refp[k])
                HIP_TRY(hipFree(pic->refp[k]));
            pic->refp[k] = nullptr, pic->refp_bytes[k] = 0;
            if (hipMalloc((void **)&pic->refp[k], need) != hipSuccess)
                return SVT_AMD_ERR_RESOURCES;
            pic->refp_bytes[k] = need;
        }
        if (bps == 1)
            hipLaunchKernelGGL(k_ep_pad<uint8_t>, dim3((stride + 255) / 256, rows), dim3(256), 0, ctx->stream, (const uint8_t *)stage[k], (int)pic->d.pitch[k], pw, ph,
                               (uint8_t *)pic->refp[k], stride, ox, oy);
        else
            hipLaunchKernelGGL(k_ep_pad<uint16_t>, dim3((stride + 255) / 256, rows), dim3(256), 0, ctx->stream, (const uint16_t *)stage[k], (int)pic->d.pitch[k], pw,
                               ph, (uint16_t *)pic->refp[k], stride, ox, oy);
        if (outs[k])
            HIP_TRY(hipMemcpyAsync(outs[k], pic->refp[k], need, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream)); /* other lanes' pictures may read the planes right away */
    ref->d_y = pic->refp[0], ref->d_cb = pic->refp[1], ref->d_cr = pic->refp[2];
    ref->strideY = w + 2 * origin_x, ref->strideC = (w >> 1) + 2 * (origin_x >> 1), ref->originX = origin_x, ref->originY = origin_y;
    ref->width = w, ref->height = h;
    return SVT_AMD_OK;
}
extern "C" int svt_amd_encdec_picture_sao(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works, const SvtAmdSaoDecisionParams *params,
                                          const uint8_t *enable, SvtAmdSaoLcuParams *lcu_params, uint8_t *out_y, uint8_t *out_cb, uint8_t *out_cr)
{
    return picture_sao<uint8_t>(ctx, pic, works, params, enable, lcu_params, out_y, out_cb, out_cr, true);
}
/* the parameter decision alone, on the picture object as it stands (deblocked: the encoder-order view; not deblocked: the picture as encoded -
 * allowEncDecMismatch pictures, whose parameters the reference decides on its un-deblocked reconstruction and signals without applying them) */
extern "C" int svt_amd_encdec_picture_sao_decide(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const void *works, const SvtAmdSaoDecisionParams *params,
                                                 const uint8_t *enable, SvtAmdSaoLcuParams *lcu_params)
{
    if (!pic || !lcu_params)
        return SVT_AMD_ERR_BAD_PARAM;
    return pic->d.bps == 1 ? picture_sao<uint8_t>(ctx, pic, (const SvtAmdLcuWork *)works, params, enable, lcu_params, nullptr, nullptr, nullptr, false)
                           : picture_sao<uint16_t>(ctx, pic, (const SvtAmdLcuWork16 *)works, params, enable, lcu_params, nullptr, nullptr, nullptr, false);
}
extern "C" int svt_amd_encdec_picture_sao16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works, const SvtAmdSaoDecisionParams *params,
                                            const uint8_t *enable, SvtAmdSaoLcuParams *lcu_params, uint16_t *out_y, uint16_t *out_cb, uint16_t *out_cr)
{
    return picture_sao<uint16_t>(ctx, pic, works, params, enable, lcu_params, out_y, out_cb, out_cr, true);
}
extern "C" int svt_amd_encdec_picture_deblock(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works, const SvtAmdLcuResult *results,
                                              const SvtAmdDeblockParams *params, uint8_t *out_y, uint8_t *out_cb, uint8_t *out_cr)
{
    return picture_deblock<uint8_t>(ctx, pic, works, results, params, out_y, out_cb, out_cr);
}
extern "C" int svt_amd_encdec_picture_deblock16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works,
                                                const SvtAmdLcuResult16 *results, const SvtAmdDeblockParams *params, uint16_t *out_y, uint16_t *out_cb,
                                                uint16_t *out_cr)
{
    return picture_deblock<uint16_t>(ctx, pic, works, results, params, out_y, out_cb, out_cr);
}

/* ---- LCUs encoded by the host: their last row / column and edge mode types enter the device picture --------------------------- */
template <typename T>
__global__ __launch_bounds__(256) void k_put_borders(EpPicture P, const typename EpTypes<T>::Border *B)
{
    const typename EpTypes<T>::Border &b = B[blockIdx.x];
    const int t = threadIdx.x;
    const int lw = min(64, (int)P.width - (int)b.lcu_x), lh = min(64, (int)P.height - (int)b.lcu_y);
    /* 0..63 bottom Y, 64..127 right Y, 128..159 / 160..191 bottom / right Cb, 192..223 / 224..255 Cr */
    const int p = t < 128 ? 0 : (t < 192 ? 1 : 2), e = p == 0 ? t : (p == 1 ? t - 128 : t - 192), n = p ? 32 : 64;
    const bool right = e >= n;
    const int i = right ? e - n : e, w = p ? lw >> 1 : lw, h = p ? lh >> 1 : lh;
    const int x0 = p ? b.lcu_x >> 1 : b.lcu_x, y0 = p ? b.lcu_y >> 1 : b.lcu_y;
    const T *src = p == 0 ? (right ? b.right_y : b.bottom_y) : p == 1 ? (right ? b.right_cb : b.bottom_cb) : (right ? b.right_cr : b.bottom_cr);
    if (i < (right ? h : w))
        ((T *)P.rec[p])[right ? (size_t)(y0 + i) * P.pitch[p] + x0 + w - 1 : (size_t)(y0 + h - 1) * P.pitch[p] + x0 + i] = src[i];
    if (t < 16 && t < (lw >> 2))
        P.mode_map[(size_t)((b.lcu_y + lh - 1) >> 2) * P.map_pitch + (b.lcu_x >> 2) + t] = b.mode_bottom[t];
    if (t >= 16 && t < 32 && t - 16 < (lh >> 2))
        P.mode_map[(size_t)((b.lcu_y >> 2) + t - 16) * P.map_pitch + ((b.lcu_x + lw - 1) >> 2)] = b.mode_right[t - 16];
}

template <typename T>
static int put_borders(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const typename EpTypes<T>::Border *borders, int n)
{
    typedef typename EpTypes<T>::Border BorderT;
    if (!ctx || !pic || !borders || n < 1 || n > 4096 || pic->d.bps != sizeof(T))
        return SVT_AMD_ERR_BAD_PARAM;
    for (int i = 0; i < n; i++)
        if (borders[i].lcu_x >= pic->d.width || borders[i].lcu_y >= pic->d.height || ((borders[i].lcu_x | borders[i].lcu_y) & 63)) {
            svt_amd_set_error("svt_amd_encdec_picture_put_borders: bad LCU %d", i);
            return SVT_AMD_ERR_BAD_PARAM;
        }
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *d = nullptr;
    int rc = svt_amd_ctx_scratch(ctx, sizeof(BorderT) * (size_t)n, &d);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(d, borders, sizeof(BorderT) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_put_borders<T>, dim3((unsigned)n), dim3(256), 0, ctx->stream, pic->d, (const BorderT *)d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ep_picture_written(ctx, pic);
}
extern "C" int svt_amd_encdec_picture_put_borders(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuBorder *borders, int n)
{
    return put_borders<uint8_t>(ctx, pic, borders, n);
}
extern "C" int svt_amd_encdec_picture_put_borders16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuBorder16 *borders, int n)
{
    return put_borders<uint16_t>(ctx, pic, borders, n);
}

/* debug: out == NULL arms the per-LCU clock sums (16 x u64 per LCU: prediction, encode, copy-out, units, wait, start, end, -, then the
 * prediction's four sub-phases),
 * a later call with a HOST buffer fetches them */
extern "C" int svt_amd_debug_encdec_profile(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, unsigned long long *out)
{
    if (!ctx || !pic)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = sizeof(unsigned long long) * 16 * (size_t)pic->nlcu;
    if (!pic->d.prof) {
        HIP_TRY(hipMalloc((void **)&pic->d.prof, bytes));
        HIP_TRY(hipMemset(pic->d.prof, 0, bytes));
    }
    if (out) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpy(out, pic->d.prof, bytes, hipMemcpyDeviceToHost));
    }
    return SVT_AMD_OK;
}
