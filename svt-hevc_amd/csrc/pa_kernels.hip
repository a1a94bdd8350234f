/*
 * pa_kernels.hip - picture-analysis statistics (include/svt_hevc_amd.h "Picture-analysis statistics") from the planes the front half
 * holds in HBM.
 *   k_pa_block_stats  ComputeBlockMeanComputeVariance (Codec/EbPictureAnalysisProcess.c:1646): one 64-thread workgroup (one wave) per LCU, a lane
 *                     per 8x8 block: four 8-byte loads (the EVEN rows of the block; ComputeSubMean8x8_SSE2_INTRIN, ASM_SSE2/EbComputeMean_Intrinsic_SSE2.c:53),
 *                     sum by v_sad_u8 against zero, sum of squares by v_dot4_u32_u8; the 16x16 / 32x32 / 64x64 levels are the reference's >> 2
 *                     averages of four children, taken across lanes with shuffles (the 8x8 blocks of a 16x16 are lanes l, l+1, l+8, l+9).
 *   k_pa_histogram    SubSampleLumaGeneratePixelIntensityHistogramBins (:3384) on the 1/16 picture: a workgroup per (region, strip of rows), 256 bins
 *                     in LDS, one global atomic per non-empty bin; the finishing touches (bins start at 1, << 4, region average) are k_pa_finish.
 *   k_sbo_ac_energy   CalculateAcEnergy (Codec/EbSourceBasedOperationsProcess.c:302; SURVEY 8f-3, an input of the mode-decision configuration): one wave per
 *                     COMPLETE LCU, a lane per 8x8 block: the block's 8x8 Hadamard sum (Compute8x8Satd_U8, C_DEFAULT/EbPictureOperators_C.c:563) in registers,
 *                     the 32x32 / 64x64 sums across lanes, AC = sum of block SATDs - (sum of the blocks' DC terms >> 2) (ComputeNxMSatdSadLCU,
 *                     Codec/EbPictureOperators.c:232).
 * Bound: HBM - 0.5 B/pel read for the block statistics (every other row; whole cache lines are fetched: 1 B/pel), 1/16 B/pel for the histograms.
 */
#include "svt_amd_internal.h"

__global__ __launch_bounds__(64) void k_pa_block_stats(const uint8_t *__restrict__ full, int pitch, int lcus_w, SvtAmdPaLcuStats *__restrict__ out)
{
    const int lcu = blockIdx.x, b = threadIdx.x; /* b: 8x8 block of the LCU, raster */
    const uint8_t *p = full + (size_t)((lcu / lcus_w) * 64 + (b >> 3) * 8) * pitch + (lcu % lcus_w) * 64 + (b & 7) * 8;
    uint32_t sum = 0, sq = 0;
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        const uint2 v = *(const uint2 *)(p + (size_t)r * pitch);
        sum = __builtin_amdgcn_sad_u8(v.x, 0u, sum);
        sum = __builtin_amdgcn_sad_u8(v.y, 0u, sum);
        sq = __builtin_amdgcn_udot4(v.x, v.x, sq, false);
        sq = __builtin_amdgcn_udot4(v.y, v.y, sq, false);
    }
    /* means with 8, means of squares with 16 fractional bits; every level above: (four children) >> 2 */
    unsigned long long m = (unsigned long long)sum << 3, s = (unsigned long long)sq << 11;
    SvtAmdPaLcuStats &o = out[lcu];
    o.y_mean[21 + b] = (uint8_t)(m >> 8), o.variance[21 + b] = (uint16_t)((s - m * m) >> 16);
    /* 16x16: lanes b, b ^ 1, b ^ 8, b ^ 9 hold its four 8x8 blocks (sums of the four are the same in each of them) */
    unsigned long long m16 = m + __shfl_xor(m, 1), s16 = s + __shfl_xor(s, 1);
    m16 = (m16 + __shfl_xor(m16, 8)) >> 2, s16 = (s16 + __shfl_xor(s16, 8)) >> 2;
    if (!(b & 9))
        o.y_mean[5 + ((b >> 4) << 2) + ((b & 7) >> 1)] = (uint8_t)(m16 >> 8), o.variance[5 + ((b >> 4) << 2) + ((b & 7) >> 1)] = (uint16_t)((s16 - m16 * m16) >> 16);
    /* 32x32: the four 16x16 of it sit at lane offsets 2 and 16 */
    unsigned long long m32 = m16 + __shfl_xor(m16, 2), s32 = s16 + __shfl_xor(s16, 2);
    m32 = (m32 + __shfl_xor(m32, 16)) >> 2, s32 = (s32 + __shfl_xor(s32, 16)) >> 2;
    if (!(b & 27))
        o.y_mean[1 + ((b >> 5) << 1) + ((b & 7) >> 2)] = (uint8_t)(m32 >> 8), o.variance[1 + ((b >> 5) << 1) + ((b & 7) >> 2)] = (uint16_t)((s32 - m32 * m32) >> 16);
    unsigned long long m64 = m32 + __shfl_xor(m32, 4), s64 = s32 + __shfl_xor(s32, 4);
    m64 = (m64 + __shfl_xor(m64, 32)) >> 2, s64 = (s64 + __shfl_xor(s64, 32)) >> 2;
    if (b == 0)
        o.y_mean[0] = (uint8_t)(m64 >> 8), o.variance[0] = (uint16_t)((s64 - m64 * m64) >> 16), o.pad = 0;
}

/* grid (strips, regions): rows [y0, y1) x columns [x0, x1) of the 1/16 picture; hist: regions x 256 counts, sums: regions x u64 */
__global__ __launch_bounds__(256) void k_pa_histogram(const uint8_t *__restrict__ six, int pitch, int width, int height, int regions_w, int regions_h,
                                                      uint32_t *__restrict__ hist, unsigned long long *__restrict__ sums)
{
    __shared__ uint32_t bins[256];
    __shared__ unsigned long long s_sum;
    const int t = threadIdx.x, region = blockIdx.y, a = region / regions_h, b = region - a * regions_h;
    const int rw = width / regions_w, rh = height / regions_h;
    const int x0 = a * rw, x1 = a == regions_w - 1 ? width : x0 + rw, y0 = b * rh, y1 = b == regions_h - 1 ? height : y0 + rh;
    bins[t] = 0;
    if (t == 0)
        s_sum = 0;
    __syncthreads();
    const int w = x1 - x0, rows = y1 - y0, strips = gridDim.x, per = (rows + strips - 1) / strips;
    const int ys = y0 + (int)blockIdx.x * per, ye = min(ys + per, y1);
    unsigned long long sum = 0;
    for (int i = t; i < (ye > ys ? (ye - ys) * w : 0); i += 256) {
        const int y = ys + i / w, x = x0 + i % w;
        const uint32_t v = six[(size_t)y * pitch + x];
        atomicAdd(&bins[v], 1u);
        sum += v;
    }
    for (int o = 32; o > 0; o >>= 1)
        sum += __shfl_xor(sum, o);
    if ((t & 63) == 0 && sum)
        atomicAdd(&s_sum, sum);
    __syncthreads();
    if (bins[t])
        atomicAdd(&hist[region * 256 + t], bins[t]);
    if (t == 0 && s_sum)
        atomicAdd(&sums[region], s_sum);
}

__global__ __launch_bounds__(256) void k_pa_finish(uint32_t *__restrict__ hist, const unsigned long long *__restrict__ sums, int width, int height, int regions_w,
                                                   int regions_h, uint8_t *__restrict__ region_average, unsigned long long *__restrict__ total)
{
    const int region = blockIdx.x, t = threadIdx.x, a = region / regions_h, b = region - a * regions_h;
    hist[region * 256 + t] = (hist[region * 256 + t] + 1u) << 4; /* bins start at 1 (InitializeBuffer_32bits ... 1) and end << 4 (:3430) */
    if (t == 0) {
        const int rw = width / regions_w, rh = height / regions_h;
        const unsigned long long w = a == regions_w - 1 ? width - a * rw : rw, h = b == regions_h - 1 ? height - b * rh : rh;
        region_average[region] = (uint8_t)((sums[region] + ((w * h) >> 1)) / (w * h));
        atomicAdd(total, sums[region] << 4);
    }
}

extern "C" int svt_amd_picture_stats(SvtAmdContext *ctx, int slot, SvtAmdPaLcuStats *out, int regions_w, int regions_h, uint32_t *histogram,
                                     uint8_t *region_average, uint64_t *sum_luma)
{
    if (!ctx || !out || slot < 0 || slot >= ctx->num_slots || regions_w < 1 || regions_h < 1 || regions_w * regions_h > 64) {
        svt_amd_set_error("svt_amd_picture_stats: bad parameter");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const DevPicture *s = &ctx->slots[slot];
    if (!s->valid) {
        svt_amd_set_error("svt_amd_picture_stats: slot %d holds no picture", slot);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    const int w = s->width, h = s->height, wl = (w + 63) / 64, hl = (h + 63) / 64, n = wl * hl, regions = regions_w * regions_h;
    if (w / 4 < regions_w || h / 4 < regions_h)
        return SVT_AMD_ERR_BAD_PARAM;
    const size_t b_out = ((size_t)n * sizeof(SvtAmdPaLcuStats) + 255) & ~(size_t)255, b_hist = (size_t)regions * 256 * 4, b_sums = (size_t)regions * 8 + 8;
    uint8_t *d = nullptr;
    int rc = svt_amd_ctx_scratch(ctx, b_out + b_hist + b_sums + 256, &d);
    if (rc)
        return rc;
    SvtAmdPaLcuStats *d_out = (SvtAmdPaLcuStats *)d;
    uint32_t *d_hist = (uint32_t *)(d + b_out);
    unsigned long long *d_sums = (unsigned long long *)(d + b_out + b_hist), *d_total = d_sums + regions;
    uint8_t *d_avg = (uint8_t *)(d_total + 1);
    HIP_TRY(hipStreamWaitEvent(ctx->stream, s->ev_ready, 0)); /* the planes may have been built on another lane */
    HIP_TRY(hipMemsetAsync(d_hist, 0, b_hist + b_sums, ctx->stream));
    hipLaunchKernelGGL(k_pa_block_stats, dim3((unsigned)n), dim3(64), 0, ctx->stream, s->full.origin, s->full.pitch, wl, d_out);
    const int strips = 16;
    hipLaunchKernelGGL(k_pa_histogram, dim3((unsigned)strips, (unsigned)regions), dim3(256), 0, ctx->stream, s->sixteenth.origin, s->sixteenth.pitch, w / 4, h / 4,
                       regions_w, regions_h, d_hist, d_sums);
    hipLaunchKernelGGL(k_pa_finish, dim3((unsigned)regions), dim3(256), 0, ctx->stream, d_hist, d_sums, w / 4, h / 4, regions_w, regions_h, d_avg, d_total);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)n * sizeof(SvtAmdPaLcuStats), hipMemcpyDeviceToHost, ctx->stream));
    if (histogram)
        HIP_TRY(hipMemcpyAsync(histogram, d_hist, b_hist, hipMemcpyDeviceToHost, ctx->stream));
    if (region_average)
        HIP_TRY(hipMemcpyAsync(region_average, d_avg, (size_t)regions, hipMemcpyDeviceToHost, ctx->stream));
    if (sum_luma)
        HIP_TRY(hipMemcpyAsync(sum_luma, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}

/* 8-point Hadamard butterflies in place (the order of the outputs differs from the reference's; only v[0] = the sum and the multiset of magnitudes matter) */
__device__ __forceinline__ void hadamard8(int *v)
{
#pragma unroll
    for (int span = 4; span > 0; span >>= 1)
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (!(i & span)) {
                const int a = v[i], b = v[i + span];
                v[i] = a + b, v[i + span] = a - b;
            }
}
/* out: [lcu][5] = the 64x64 then the four 32x32 (raster) of every LCU; incomplete LCUs carry the reference's "not computed" value (:351) */
__global__ __launch_bounds__(64) void k_sbo_ac_energy(const uint8_t *__restrict__ full, int pitch, int width, int height, int lcus_w,
                                                      unsigned long long *__restrict__ out)
{
    const int lcu = blockIdx.x, b = threadIdx.x, lx = (lcu % lcus_w) * 64, ly = (lcu / lcus_w) * 64;
    if (lx + 64 > width || ly + 64 > height) {
        if (b < 5)
            out[(size_t)lcu * 5 + b] = 100000000ull;
        return;
    }
    const uint8_t *p = full + (size_t)(ly + (b >> 3) * 8) * pitch + lx + (b & 7) * 8;
    int m[8][8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint2 v = *(const uint2 *)(p + (size_t)r * pitch);
#pragma unroll
        for (int c = 0; c < 4; c++)
            m[r][c] = (v.x >> (8 * c)) & 255, m[r][4 + c] = (v.y >> (8 * c)) & 255;
        hadamard8(m[r]);
    }
    uint32_t satd = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        int col[8];
#pragma unroll
        for (int r = 0; r < 8; r++)
            col[r] = m[r][c];
        hadamard8(col);
#pragma unroll
        for (int r = 0; r < 8; r++)
            satd += (uint32_t)abs(col[r]), m[r][c] = col[r];
    }
    uint32_t s = (satd + 2) >> 2, dc = (uint32_t)m[0][0]; /* Compute8x8Satd_U8: the block's rounded sum; *dcValue += m2[0][0] */
    /* the 32x32 the lane's block belongs to: lanes that differ in bits 0, 1 (x) and 3, 4 (y) */
    s += __shfl_xor(s, 1), dc += __shfl_xor(dc, 1);
    s += __shfl_xor(s, 2), dc += __shfl_xor(dc, 2);
    s += __shfl_xor(s, 8), dc += __shfl_xor(dc, 8);
    s += __shfl_xor(s, 16), dc += __shfl_xor(dc, 16);
    if (!(b & 27))
        out[(size_t)lcu * 5 + 1 + ((b >> 5) << 1) + ((b & 7) >> 2)] = (unsigned long long)s - (dc >> 2);
    s += __shfl_xor(s, 4), dc += __shfl_xor(dc, 4);
    s += __shfl_xor(s, 32), dc += __shfl_xor(dc, 32);
    if (b == 0)
        out[(size_t)lcu * 5] = (unsigned long long)s - (dc >> 2);
}

extern "C" int svt_amd_picture_ac_energy(SvtAmdContext *ctx, int slot, uint64_t *out)
{
    if (!ctx || !out || slot < 0 || slot >= ctx->num_slots) {
        svt_amd_set_error("svt_amd_picture_ac_energy: bad parameter");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const DevPicture *s = &ctx->slots[slot];
    if (!s->valid) {
        svt_amd_set_error("svt_amd_picture_ac_energy: slot %d holds no picture", slot);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    const int wl = (s->width + 63) / 64, hl = (s->height + 63) / 64, n = wl * hl;
    uint8_t *d = nullptr;
    int rc = svt_amd_ctx_scratch(ctx, (size_t)n * 40, &d);
    if (rc)
        return rc;
    HIP_TRY(hipStreamWaitEvent(ctx->stream, s->ev_ready, 0));
    hipLaunchKernelGGL(k_sbo_ac_energy, dim3((unsigned)n), dim3(64), 0, ctx->stream, s->full.origin, s->full.pitch, s->width, s->height, wl, (unsigned long long *)d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d, (size_t)n * 40, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}
