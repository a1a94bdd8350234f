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
 * (sizeof(PrepJobDev) is mirrored in context.hip for the descriptor allocation.)
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

/* ------------------------------------------------------------------------------------------------
 * Fused preparation: ONE launch builds all six planes of every picture of a batch straight from the
 * raw luma (clamped coordinates = replicated borders), so nothing is written and read back in between.
 *   section 0  full-resolution column strips: a thread owns 4 samples x STRIP rows and slides a 4-row
 *              window of F and B down the strip: per row it emits the padded F dword, the B dword,
 *              and one row later H (vertical filter of F) and J (vertical filter of B).
 *   section 1  1/4 plane, section 2  1/16 plane (point decimation), 4 samples x 8 rows per thread.
 * Traffic per 1080p picture: 2.1 MB read (+ halo) and ~12 MB written.
 * ------------------------------------------------------------------------------------------------ */
#define STRIP 16

struct PrepJobDev {
    const uint8_t *raw;
    int32_t raw_stride, w, h;
    uint8_t *full, *quarter, *sixteenth, *B, *H, *J;
    int32_t pitch_full, pitch_quarter, pitch_sixteenth;
};

__device__ __forceinline__ uint32_t vfilt4(uint32_t m2, uint32_t m1, uint32_t c0, uint32_t p1)
{
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int s = 8 * i;
        v |= round_shift_clip(-2 * (int)((m2 >> s) & 255) + 18 * (int)((m1 >> s) & 255) + 18 * (int)((c0 >> s) & 255) -
                              2 * (int)((p1 >> s) & 255) + 16) << s;
    }
    return v;
}

/* F(x0-2 .. x0+4) of clamped row `row` (raw picture row pointer); fast = aligned interior */
__device__ __forceinline__ void load7(const uint8_t *row, int x0, int w, bool fast, int (&a)[7], uint32_t &fd)
{
    if (fast) {
        const uint32_t d0 = *(const uint32_t *)(row + x0 - 4), d1 = *(const uint32_t *)(row + x0),
                       d2 = *(const uint32_t *)(row + x0 + 4);
        a[0] = (d0 >> 16) & 255, a[1] = d0 >> 24;
        a[2] = d1 & 255, a[3] = (d1 >> 8) & 255, a[4] = (d1 >> 16) & 255, a[5] = d1 >> 24;
        a[6] = d2 & 255;
        fd = d1;
    } else {
#pragma unroll
        for (int i = 0; i < 7; i++)
            a[i] = row[clampi(x0 - 2 + i, 0, w - 1)];
        fd = (uint32_t)a[2] | ((uint32_t)a[3] << 8) | ((uint32_t)a[4] << 16) | ((uint32_t)a[5] << 24);
    }
}

__global__ __launch_bounds__(256) void k_prep_fused(const PrepJobDev *__restrict__ jobs, int b0, int b1)
{
    const PrepJobDev J = jobs[blockIdx.y];
    const int w = J.w, h = J.h;
    int blk = (int)blockIdx.x;
    if (blk < b0) {
        const int pad = SVT_AMD_PAD_FULL;
        const int cols4 = (w + 2 * pad) >> 2, strips = (h + 2 * pad + STRIP - 1) / STRIP;
        const int g = blk * 256 + (int)threadIdx.x;
        if (g >= cols4 * strips)
            return;
        const int strip = g / cols4, cx = g - strip * cols4;
        const int x0 = -pad + 4 * cx, ys = -pad + STRIP * strip, ye = min(ys + STRIP, h + pad);
        const bool fast = x0 >= 4 && x0 + 8 <= w && ((((uintptr_t)J.raw) | (uint32_t)J.raw_stride) & 3) == 0;
        uint32_t f[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0}; /* rows r-3 .. r */
        const ptrdiff_t pf = J.pitch_full;
        for (int r = ys - 2; r <= ye; r++) {
            const uint8_t *row = J.raw + (size_t)clampi(r, 0, h - 1) * J.raw_stride;
            int a[7];
            uint32_t fd, vb = 0;
            load7(row, x0, w, fast, a, fd);
#pragma unroll
            for (int i = 0; i < 4; i++)
                vb |= round_shift_clip(-2 * a[i] + 18 * a[i + 1] + 18 * a[i + 2] - 2 * a[i + 3] + 16) << (8 * i);
            f[0] = f[1], f[1] = f[2], f[2] = f[3], f[3] = fd;
            b[0] = b[1], b[1] = b[2], b[2] = b[3], b[3] = vb;
            if (r >= ys && r < ye) {
                *(uint32_t *)(J.full + r * pf + x0) = fd;
                *(uint32_t *)(J.B + r * pf + x0) = vb;
            }
            const int y = r - 1; /* window now holds rows y-2 .. y+1 */
            if (y >= ys && y < ye && r >= ys + 1) {
                *(uint32_t *)(J.H + y * pf + x0) = vfilt4(f[0], f[1], f[2], f[3]);
                *(uint32_t *)(J.J + y * pf + x0) = vfilt4(b[0], b[1], b[2], b[3]);
            }
        }
        return;
    }
    /* decimated planes */
    const bool q = blk < b0 + b1;
    blk -= q ? b0 : b0 + b1;
    const int step = q ? 2 : 4, pad = q ? SVT_AMD_PAD_QUARTER : SVT_AMD_PAD_SIXTEENTH;
    const int dw = w / step, dh = h / step;
    uint8_t *dst = q ? J.quarter : J.sixteenth;
    const ptrdiff_t pitch = q ? J.pitch_quarter : J.pitch_sixteenth;
    const int cols4 = (dw + 2 * pad + 3) >> 2, strips = (dh + 2 * pad + 7) / 8;
    const int g = blk * 256 + (int)threadIdx.x;
    if (g >= cols4 * strips)
        return;
    const int strip = g / cols4, cx = g - strip * cols4;
    const int x0 = -pad + 4 * cx, ys = -pad + 8 * strip, ye = min(ys + 8, dh + pad);
    const bool fast = x0 >= 0 && x0 + 4 <= dw && ((((uintptr_t)J.raw) | (uint32_t)J.raw_stride) & (q ? 7 : 15)) == 0;
    for (int y = ys; y < ye; y++) {
        const uint8_t *row = J.raw + (size_t)(clampi(y, 0, dh - 1) * step) * J.raw_stride;
        uint32_t v;
        if (fast && q) {
            const uint2 d = *(const uint2 *)(row + x0 * 2);
            v = __builtin_amdgcn_perm(d.y, d.x, 0x06040200u);
        } else if (fast) {
            const uint4 d = *(const uint4 *)(row + x0 * 4);
            v = (d.x & 255) | ((d.y & 255) << 8) | ((d.z & 255) << 16) | ((d.w & 255) << 24);
        } else {
            v = 0;
#pragma unroll
            for (int i = 0; i < 4; i++)
                v |= (uint32_t)row[clampi(x0 + i, 0, dw - 1) * step] << (8 * i);
        }
        *(uint32_t *)(dst + y * pitch + x0) = v;
    }
}

static_assert(sizeof(PrepJobDev) <= 128, "context.hip reserves 128 bytes per prep descriptor");

int svt_amd_launch_prep_batch(SvtAmdContext *ctx, DevPicture *const *pics, const uint8_t *const *d_luma, uint32_t stride,
                              int n)
{
    static thread_local PrepJobDev host[SVT_AMD_MAX_BATCH];
    const int w = pics[0]->width, h = pics[0]->height;
    for (int i = 0; i < n; i++) {
        const DevPicture *p = pics[i];
        PrepJobDev &j = host[i];
        j.raw = d_luma[i], j.raw_stride = (int)stride, j.w = w, j.h = h;
        j.full = p->full.origin, j.quarter = p->quarter.origin, j.sixteenth = p->sixteenth.origin;
        j.B = p->hp_b.origin, j.H = p->hp_h.origin, j.J = p->hp_j.origin;
        j.pitch_full = p->full.pitch, j.pitch_quarter = p->quarter.pitch, j.pitch_sixteenth = p->sixteenth.pitch;
    }
    const int padf = SVT_AMD_PAD_FULL;
    const int n0 = ((w + 2 * padf) >> 2) * ((h + 2 * padf + STRIP - 1) / STRIP);
    const int n1 = (((w >> 1) + 2 * SVT_AMD_PAD_QUARTER + 3) >> 2) * (((h >> 1) + 2 * SVT_AMD_PAD_QUARTER + 7) / 8);
    const int n2 = (((w >> 2) + 2 * SVT_AMD_PAD_SIXTEENTH + 3) >> 2) * (((h >> 2) + 2 * SVT_AMD_PAD_SIXTEENTH + 7) / 8);
    const int b0 = (n0 + 255) / 256, b1 = (n1 + 255) / 256, b2 = (n2 + 255) / 256;
    int rc = svt_amd_stamp_begin(ctx, KC_PREP);
    if (rc)
        return rc;
    {
        const int rcd = svt_amd_upload_descriptors(ctx, ctx->d_prep_jobs, host, sizeof(PrepJobDev) * (size_t)n);
        if (rcd)
            return rcd;
    }
    hipLaunchKernelGGL(k_prep_fused, dim3(b0 + b1 + b2, n), dim3(256), 0, ctx->stream, (const PrepJobDev *)ctx->d_prep_jobs, b0, b1);
    HIP_TRY(hipGetLastError());
    return svt_amd_stamp_end(ctx);
}

int svt_amd_launch_prep(SvtAmdContext *ctx, DevPicture *pic, const uint8_t *d_luma, uint32_t stride)
{
    return svt_amd_launch_prep_batch(ctx, &pic, &d_luma, stride, 1);
}
