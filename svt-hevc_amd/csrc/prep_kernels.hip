/*
 * Picture preparation kernels (once per uploaded picture):
 *   build_plane     padded full plane, and the 1/4 and 1/16 point-decimated planes with
 *                   their own replicated borders - all three read the raw luma directly
 *                   with clamped coordinates, so there is no pad->decimate->pad chain.
 *                   Reference: GeneratePadding (Codec/EbMcp.c:1017), Decimation2D
 *                   (Codec/EbPictureAnalysisProcess.c:173), DecimateInputPicture (:4139).
 *   halfpel_bh/j    AVC-style half-pel planes b, h and j, filter {-2,18,18,-2},(+16)>>5,
 *                   clip (AvcStyleLumaInterpolationFilterHorizontal/Vertical,
 *                   C_DEFAULT/EbAvcStyleMcp_C.c:34,62; call pattern of
 *                   EbHevcInterpolateSearchRegionAVC, Codec/EbMotionEstimation.c:645-727).
 *
 * Streaming, HBM-bound: one thread produces 4 horizontally adjacent samples and
 * writes them as one dword; a wave covers 256 contiguous bytes of a row.
 */
#include "svt_amd_internal.h"

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* clip((sum + 16) >> 5) to [0,255].  Written as max(sum,0) -> logical shift -> umin on purpose:
 * the natural form clamp(sum >> 5, 0, 255) is pattern-matched by hipcc 7.2 into
 * v_ashr_pk_u8_i32 pairs whose upper 16 result bits are not zero on gfx950 when an input
 * is negative, which corrupted the two neighbouring samples (found by the parity test). */
__device__ __forceinline__ uint32_t round_shift_clip(int sum_plus_16)
{
    const uint32_t u = (uint32_t)(sum_plus_16 < 0 ? 0 : sum_plus_16) >> 5;
    return u > 255u ? 255u : u;
}

/* dst(x,y) = src[clamp(y) * step][clamp(x) * step] for x in [-pad, w+pad), y in [-pad, h+pad) */
__global__ __launch_bounds__(256) void k_build_plane(uint8_t *__restrict__ dst, int pitch, int w, int h, int pad,
                                                     const uint8_t *__restrict__ src, int src_stride, int step)
{
    const int x0 = -pad + 4 * (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int y = -pad + (int)blockIdx.y;
    if (x0 >= w + pad)
        return;
    const uint8_t *row = src + (size_t)(clampi(y, 0, h - 1) * step) * src_stride;
    uint32_t v;
    if (step == 1 && x0 >= 0 && x0 + 3 < w && ((((uintptr_t)(row + x0)) & 3) == 0)) {
        v = *(const uint32_t *)(row + x0);
    } else {
        v = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
            v |= (uint32_t)row[clampi(x0 + i, 0, w - 1) * step] << (8 * i);
    }
    *(uint32_t *)(dst + (ptrdiff_t)y * pitch + x0) = v;
}

/* B(x,y) and H(x,y) from the padded full plane F. */
__global__ __launch_bounds__(256) void k_halfpel_bh(const uint8_t *__restrict__ F, uint8_t *__restrict__ B,
                                                    uint8_t *__restrict__ H, int pitch, int w, int h, int pad)
{
    const int x0 = -pad + 4 * (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int y = -pad + (int)blockIdx.y;
    if (x0 >= w + pad)
        return;
    const uint8_t *r = F + (ptrdiff_t)y * pitch + x0;
    uint32_t vb = 0, vh = 0;
    int a[7]; /* F(x0-2 .. x0+4) */
#pragma unroll
    for (int i = 0; i < 7; i++)
        a[i] = r[i - 2];
#pragma unroll
    for (int i = 0; i < 4; i++)
        vb |= round_shift_clip(-2 * a[i] + 18 * a[i + 1] + 18 * a[i + 2] - 2 * a[i + 3] + 16) << (8 * i);
    const uint32_t m2 = *(const uint32_t *)(r - 2 * pitch), m1 = *(const uint32_t *)(r - pitch),
                   c0 = *(const uint32_t *)r, p1 = *(const uint32_t *)(r + pitch);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int s = 8 * i;
        vh |= round_shift_clip(-2 * (int)((m2 >> s) & 255) + 18 * (int)((m1 >> s) & 255) + 18 * (int)((c0 >> s) & 255) -
                               2 * (int)((p1 >> s) & 255) + 16) << s;
    }
    *(uint32_t *)(B + (ptrdiff_t)y * pitch + x0) = vb;
    *(uint32_t *)(H + (ptrdiff_t)y * pitch + x0) = vh;
}

/* J(x,y) = vertical filter over B rows y-2..y+1 */
__global__ __launch_bounds__(256) void k_halfpel_j(const uint8_t *__restrict__ B, uint8_t *__restrict__ J, int pitch,
                                                   int w, int h, int pad)
{
    const int x0 = -pad + 4 * (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int y = -pad + 2 + (int)blockIdx.y; /* rows [-pad+2, h+pad-1) */
    if (x0 >= w + pad)
        return;
    const uint8_t *r = B + (ptrdiff_t)y * pitch + x0;
    const uint32_t m2 = *(const uint32_t *)(r - 2 * pitch), m1 = *(const uint32_t *)(r - pitch),
                   c0 = *(const uint32_t *)r, p1 = *(const uint32_t *)(r + pitch);
    uint32_t vj = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int s = 8 * i;
        vj |= round_shift_clip(-2 * (int)((m2 >> s) & 255) + 18 * (int)((m1 >> s) & 255) + 18 * (int)((c0 >> s) & 255) -
                               2 * (int)((p1 >> s) & 255) + 16) << s;
    }
    *(uint32_t *)(J + (ptrdiff_t)y * pitch + x0) = vj;
}

static dim3 grid_for(int w, int h, int pad, int rows_trim)
{
    const int cols4 = (w + 2 * pad + 3) / 4;
    return dim3((unsigned)((cols4 + 255) / 256), (unsigned)(h + 2 * pad - rows_trim), 1);
}

int svt_amd_launch_prep(SvtAmdContext *ctx, DevPicture *pic, const uint8_t *d_luma, uint32_t stride)
{
    const int w = pic->width, h = pic->height;
    int rc = svt_amd_stamp_begin(ctx, KC_PREP);
    if (rc)
        return rc;
    hipLaunchKernelGGL(k_build_plane, grid_for(w, h, SVT_AMD_PAD_FULL, 0), dim3(256), 0, ctx->stream,
                       pic->full.origin, pic->full.pitch, w, h, SVT_AMD_PAD_FULL, d_luma, (int)stride, 1);
    hipLaunchKernelGGL(k_build_plane, grid_for(w >> 1, h >> 1, SVT_AMD_PAD_QUARTER, 0), dim3(256), 0, ctx->stream,
                       pic->quarter.origin, pic->quarter.pitch, w >> 1, h >> 1, SVT_AMD_PAD_QUARTER, d_luma,
                       (int)stride, 2);
    hipLaunchKernelGGL(k_build_plane, grid_for(w >> 2, h >> 2, SVT_AMD_PAD_SIXTEENTH, 0), dim3(256), 0, ctx->stream,
                       pic->sixteenth.origin, pic->sixteenth.pitch, w >> 2, h >> 2, SVT_AMD_PAD_SIXTEENTH, d_luma,
                       (int)stride, 4);
    hipLaunchKernelGGL(k_halfpel_bh, grid_for(w, h, SVT_AMD_PAD_FULL, 0), dim3(256), 0, ctx->stream,
                       pic->full.origin, pic->hp_b.origin, pic->hp_h.origin, pic->full.pitch, w, h, SVT_AMD_PAD_FULL);
    hipLaunchKernelGGL(k_halfpel_j, grid_for(w, h, SVT_AMD_PAD_FULL, 3), dim3(256), 0, ctx->stream,
                       pic->hp_b.origin, pic->hp_j.origin, pic->full.pitch, w, h, SVT_AMD_PAD_FULL);
    HIP_TRY(hipGetLastError());
    return svt_amd_stamp_end(ctx);
}
