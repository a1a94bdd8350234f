/*
 * Coefficient rate estimation on the device (EncDec row "TuEstimateCoeffBitsLuma,
 * EstimateQuantizedCoefficients_*"): EstimateQuantizedCoefficients_Lossy
 * (Codec/EbCoeffEstimation_Intrinsic.c:1415-1842), the estimator behind TuEstimateCoeffBits* and the
 * PM-core quantiser.
 *
 * The reference walks the 4x4 sub-blocks of a TU serially, but in this ("lossy") estimator no state is carried from
 * one sub-block to the next (fixed greater-1 context set, rice parameter 0, prevCsbf = 0 significance contexts), so
 * the device form gives every sub-block its own lane: a 32x32 TU is one wavefront, smaller TUs share a wave
 * (64 / sub-blocks TUs per wave).  A lane loads its 16 coefficients in scan order, a ballot finds the last
 * significant sub-block, every lane prices its sub-block with a fully unrolled 16-position loop (no arrays), a
 * segmented shuffle adds the TU up.  Scan / context tables are generated from their H.265 definitions at load.
 * Bound: HBM (2 bytes per coefficient in, 8 bytes per TU out); ~150 integer ops per coefficient.
 */
#include "rate_device.h"

/* lg = log2(size) in 2..5, S = sub-blocks per TU = lanes per TU */
__global__ __launch_bounds__(256) void k_coeff_bits(const SvtAmdCabacCost *__restrict__ cost_p, const int16_t *__restrict__ coeff, uint32_t stride, size_t block_pitch,
                                                    const SvtAmdTuInfo *__restrict__ info, unsigned long long *__restrict__ out,
                                                    uint32_t nblocks, int lg)
{
    const int S = lg == 2 ? 1 : 1 << (2 * (lg - 2)), tpw = 64 / S;
    const int lane = threadIdx.x & 63, sub = lane & (S - 1);
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint32_t b = wave * tpw + lane / S;
    const bool live = b < nblocks;
    SvtAmdTuInfo ti = {0, 1, 0, 0, 0};
    if (live)
        ti = info[b];
    const uint32_t bits = coeff_bits_lanes(*cost_p, coeff + (size_t)(live ? b : 0) * block_pitch, stride, lg, ti, live, lane, sub);
    if (live && sub == 0)
        out[b] = (unsigned long long)bits << 10;
}

static int launch_rate(hipStream_t st, const SvtAmdCabacCost *d_cost, uint32_t size, const int16_t *d_coeff, uint32_t stride, size_t pitch,
                       const SvtAmdTuInfo *d_info, unsigned long long *d_bits, uint32_t n)
{
    const int lg = size == 4 ? 2 : size == 8 ? 3 : size == 16 ? 4 : size == 32 ? 5 : 0;
    if (!lg || !n)
        return SVT_AMD_ERR_BAD_PARAM;
    const int S = lg == 2 ? 1 : 1 << (2 * (lg - 2)), tpw = 64 / S;
    const uint32_t waves = (n + tpw - 1) / tpw;
    hipLaunchKernelGGL(k_coeff_bits, dim3((waves + 3) / 4), dim3(256), 0, st, d_cost, d_coeff, stride, pitch, d_info, d_bits, n, lg);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

extern "C" int svt_amd_coeff_bits_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, uint32_t size,
                                        const int16_t *d_coeff, const SvtAmdTuInfo *d_info, uint64_t *d_bits,
                                        uint32_t nblocks)
{
    if (!ctx || !cost || !d_coeff || !d_info || !d_bits)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rate_upload_tables(ctx, cost);
    if (rc)
        return rc;
    return launch_rate(ctx->stream, (const SvtAmdCabacCost *)ctx->d_cabac_cost, size, d_coeff, size, (size_t)size * size, d_info, (unsigned long long *)d_bits, nblocks);
}

extern "C" int svt_amd_EstimateQuantizedCoefficients_Lossy(SvtAmdCabacCost *CabacCost, void *cabacEncodeCtxPtr, uint32_t size,
                                                           uint32_t type, uint32_t intraLumaMode, uint32_t intraChromaMode,
                                                           int16_t *coeffBufferPtr, const uint32_t coeffStride,
                                                           uint32_t componentType, uint32_t numNonZeroCoeffs,
                                                           uint64_t *coeffBitsLong)
{
    (void)cabacEncodeCtxPtr;
    if (!CabacCost || !coeffBufferPtr || !coeffBitsLong || !numNonZeroCoeffs)
        return 1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || rate_tables_once(dev))
        return 1;
    SvtAmdTuInfo hi = {numNonZeroCoeffs, (uint8_t)type, (uint8_t)intraLumaMode, (uint8_t)intraChromaMode, (uint8_t)componentType};
    DBuf c(coeffBufferPtr, span(coeffStride, size, size) * 2), i(&hi, sizeof(hi)), o(nullptr, 8, false), k(CabacCost, sizeof(*CabacCost));
    if (!(c.ok && i.ok && o.ok && k.ok))
        return 1;
    if (launch_rate(0, (const SvtAmdCabacCost *)k.d, size, (const int16_t *)c.d, coeffStride, 0, (const SvtAmdTuInfo *)i.d, (unsigned long long *)o.d, 1))
        return 1;
    unsigned long long v = 0;
    if (!finish("EstimateQuantizedCoefficients_Lossy") || !o.download(&v, 8))
        return 1;
    *coeffBitsLong += v;
    return 0;
}

/* ---- context-updating estimator, standalone (EstimateQuantizedCoefficients_generic_Update) -----------------------------
 * One workgroup (one wave) per CHAIN of `chain_len` consecutive blocks that share one model, walked in order like the mode
 * decision threads the model through the units of a candidate: the wave builds a block's significance maps together, lane 0
 * walks it.  ctx: [nchains][136] words in CoeffCtxtMdl_t's order, updated in place. */
__global__ __launch_bounds__(64) void k_coeff_bits_update(const int16_t *__restrict__ coeff, uint32_t stride, size_t block_pitch,
                                                          const SvtAmdTuInfo *__restrict__ info, uint32_t *__restrict__ ctx,
                                                          unsigned long long *__restrict__ out, uint32_t nblocks, uint32_t chain_len, int lg)
{
    __shared__ uint8_t M[RATE_CTX_WORDS];
    __shared__ uint16_t sigm[64], absC[16];
    const int t = threadIdx.x;
    uint32_t *cw = ctx + (size_t)blockIdx.x * RATE_CTX_WORDS;
    for (int i = t; i < RATE_CTX_WORDS; i += 64)
        M[i] = (uint8_t)cw[i];
    __syncthreads();
    for (uint32_t k = 0; k < chain_len; k++) {
        const uint32_t b = blockIdx.x * chain_len + k;
        if (b >= nblocks)
            break;
        const SvtAmdTuInfo ti = info[b];
        const int16_t *p0 = coeff + (size_t)b * block_pitch;
        if (ti.num_nonzero)
            rate_update_sigmaps(sigm, p0, stride, lg, ti, t, 64);
        __syncthreads();
        if (t == 0)
            out[b] = ti.num_nonzero ? coeff_bits_update_walk(M, sigm, absC, p0, stride, lg, ti) : 0ull;
        __syncthreads();
    }
    for (int i = t; i < RATE_CTX_WORDS; i += 64)
        cw[i] = M[i];
}

extern "C" int svt_amd_coeff_bits_update_batch(SvtAmdContext *ctx, uint32_t size, const int16_t *d_coeff, const SvtAmdTuInfo *d_info,
                                               uint32_t *d_ctx_models, uint64_t *d_bits, uint32_t nblocks, uint32_t chain_len)
{
    const int lg = size == 4 ? 2 : size == 8 ? 3 : size == 16 ? 4 : size == 32 ? 5 : 0;
    if (!ctx || !d_coeff || !d_info || !d_ctx_models || !d_bits || !lg || !nblocks || !chain_len)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rate_tables_once(ctx->device);
    if (rc)
        return rc;
    hipLaunchKernelGGL(k_coeff_bits_update, dim3((nblocks + chain_len - 1) / chain_len), dim3(64), 0, ctx->stream, d_coeff, size,
                       (size_t)size * size, d_info, d_ctx_models, (unsigned long long *)d_bits, nblocks, chain_len, lg);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* LEAF: slot of the table EstimateQuantizedCoefficientsUpdate (Codec/EbEntropyCoding.h:387), reference signature */
extern "C" int svt_amd_EstimateQuantizedCoefficients_Update(uint32_t *updatedCoeffCtxModel, SvtAmdCabacCost *CabacCost, void *cabacEncodeCtxPtr,
                                                            uint32_t size, uint32_t type, uint32_t intraLumaMode, uint32_t intraChromaMode,
                                                            int16_t *coeffBufferPtr, const uint32_t coeffStride, uint32_t componentType,
                                                            uint32_t numNonZeroCoeffs, uint64_t *coeffBitsLong)
{
    (void)CabacCost;
    (void)cabacEncodeCtxPtr;
    const int lg = size == 4 ? 2 : size == 8 ? 3 : size == 16 ? 4 : size == 32 ? 5 : 0;
    if (!updatedCoeffCtxModel || !coeffBufferPtr || !coeffBitsLong || !numNonZeroCoeffs || !lg)
        return 1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || rate_tables_once(dev))
        return 1;
    SvtAmdTuInfo hi = {numNonZeroCoeffs, (uint8_t)type, (uint8_t)intraLumaMode, (uint8_t)intraChromaMode, (uint8_t)componentType};
    DBuf c(coeffBufferPtr, span(coeffStride, size, size) * 2), i(&hi, sizeof(hi)), o(nullptr, 8, false),
        m(updatedCoeffCtxModel, RATE_CTX_WORDS * 4);
    if (!(c.ok && i.ok && o.ok && m.ok))
        return 1;
    hipLaunchKernelGGL(k_coeff_bits_update, dim3(1), dim3(64), 0, 0, (const int16_t *)c.d, coeffStride, (size_t)0, (const SvtAmdTuInfo *)i.d,
                       (uint32_t *)m.d, (unsigned long long *)o.d, 1u, 1u, lg);
    unsigned long long v = 0;
    if (!finish("EstimateQuantizedCoefficients_Update") || !o.download(&v, 8) || !m.download(updatedCoeffCtxModel, RATE_CTX_WORDS * 4))
        return 1;
    *coeffBitsLong += v;
    return 0;
}
