/*
 * Luma and chroma full loops of mode-decision candidates, fused (EncDec row "PerformFullLoop, ProductFullLoop ...").
 *
 * Replaces ProductFullLoop (Codec/EbFullLoop.c:185-446) for the presets' common configuration (no RDOQ / PM-core,
 * coefficient-domain distortion, no CABAC-context update).  One workgroup per candidate CU, transform units one after
 * the other (one, or four 32x32 for a 64x64 CU); everything between the residual and the cost decision stays on chip:
 *   residual -> LDS -> forward "Estimate" DCT (two passes in LDS, txfm_device.h)
 *            -> quantise / inverse-quantise the (T >> pf) area, count non-zeros, coefficient-domain distortions
 *            -> coefficient bits (one lane per 4x4 sub-block, rate_device.h) -> cbf / cost decision (one thread).
 * Only the quantised and reconstructed coefficients and a 64-byte result record leave the workgroup.
 * HBM traffic per TU: 2 B/sample in, 4 B/sample out (area only) - the five-kernel composition moves 14 B/sample.
 */
#include "txfm_device.h"
#include "rate_device.h"
#include <cstring>

struct FlShared {
    int16_t q[32 * 32];       /* quantised coefficients of the current TU, row pitch N */
    unsigned nz, res, pred;   /* per-TU accumulators */
    uint32_t bits;
};

template <int N>
__global__ __launch_bounds__(TX_THREADS) void k_full_loop_luma(const SvtAmdFullLoopIn *__restrict__ in_all,
                                                              const int16_t *__restrict__ residual,
                                                              int16_t *__restrict__ quant, int16_t *__restrict__ recon,
                                                              SvtAmdFullLoopOut *__restrict__ out_all, int shift1, int shift2,
                                                              int wrap_levels)
{
    __shared__ TxShared<N> X;
    __shared__ FlShared F;
    const SvtAmdFullLoopIn in = in_all[blockIdx.x];
    const int size = (int)in.size, T = size == 64 ? 32 : size;
    if (T != N)
        return; /* this launch serves the other transform sizes */
    const int t = threadIdx.x, ntu = size == 64 ? 4 : 1, pitch = size;
    const size_t base = (size_t)blockIdx.x * 4096;
    const int area = N >> in.pf_mode;
    for (int i = t; i < 32 * 32; i += TX_THREADS)
        (&X.T[0][0])[i] = (&c_T32[0][0])[i];
    /* ProductUnifiedQuantizeInvQuantizeMd (EbFullLoop.c:98-113) */
    const int qpRem = (int)(in.qp % 6), qpPer = (int)(in.qp / 6);
    const uint32_t QF = qpRem == 0 ? 26214u : qpRem == 1 ? 23302u : qpRem == 2 ? 20560u : qpRem == 3 ? 18396u : qpRem == 4 ? 16384u : 14564u;
    const int FFv = qpRem == 0 ? 40 : qpRem == 1 ? 45 : qpRem == 2 ? 51 : qpRem == 3 ? 57 : qpRem == 4 ? 64 : 72;
    constexpr int LG = N == 32 ? 5 : N == 16 ? 4 : 3;
    const int tshift = 7 - LG, shiftedQBits = 14 + qpPer + tshift;
    const uint32_t q_offset = ((in.slice_type == 2 || in.slice_type == 3) ? 171u : 85u) << (shiftedQBits - 9);
    const int shiftedFFunc = qpPer > 8 ? FFv << (qpPer - 2) : FFv << qpPer;
    const int shiftNum = qpPer > 8 ? 20 - 14 - tshift - 2 : 20 - 14 - tshift;
    const int iq_offset = 1 << (shiftNum - 1);

    uint32_t ycbf = in.ycbf;
    unsigned long long bits_acc = in.coeff_bits, d0_acc = size == 64 ? in.dist[0] : 0, d1_acc = size == 64 ? in.dist[1] : 0;
    SvtAmdFullLoopOut o;
    for (int k = 0; k < 5; k++)
        o.nz[k] = 0;
    for (int k = 0; k < 4; k++)
        o.ydc[k] = 0, o.cand_nz[k] = 0;

    for (int tu = 0; tu < ntu; tu++) {
        const int off = size == 64 ? ((tu & 1) << 5) + (tu > 1 ? 32 * 64 : 0) : 0;
        constexpr int GB = TxShared<N>::GB;
        for (int i = t; i < GB * N * N; i += TX_THREADS)
            (&X.io[0][0])[i] = i < N * N ? residual[base + off + (i / N) * pitch + (i % N)] : (int16_t)0;
        if (t == 0)
            F.nz = 0, F.res = 0, F.pred = 0, F.bits = 0;
        __syncthreads();
        fwd_pass<N, false>(X, shift1, wrap_levels, nullptr, 1, t);
        fwd_pass<N, false>(X, shift2, wrap_levels, nullptr, 1, t);
        /* QuantizeInvQuantize (C_DEFAULT/EbTransforms_C.c:89) over the area + the two coefficient-domain sums */
        unsigned nz = 0, res = 0, pred = 0;
        for (int i = t; i < area * area; i += TX_THREADS) {
            const int r = i / area, c = i - r * area;
            const int v = X.io[0][r * N + c], sign = v < 0 ? -1 : 1;
            int tq = abs(v);
            tq = (int)((uint32_t)tq * QF);
            tq = (int)((uint32_t)tq + q_offset);
            tq >>= shiftedQBits;
            const int qv = clip16i(sign * tq);
            const int rv = clip16i(((qv * shiftedFFunc) + iq_offset) >> shiftNum);
            F.q[r * N + c] = (int16_t)qv;
            quant[base + off + r * pitch + c] = (int16_t)qv;
            recon[base + off + r * pitch + c] = (int16_t)rv;
            nz += qv != 0;
            const int16_t d = (int16_t)(v - rv);
            res += (unsigned)(d * d);
            pred += (unsigned)(v * v);
        }
        for (int s = 32; s > 0; s >>= 1)
            nz += __shfl_xor(nz, s), res += __shfl_xor(res, s), pred += __shfl_xor(pred, s);
        if ((t & 63) == 0) {
            atomicAdd(&F.nz, nz);
            atomicAdd(&F.res, res);
            atomicAdd(&F.pred, pred);
        }
        __syncthreads();
        const unsigned tnz = F.nz;
        /* TuEstimateCoeffBitsLuma: wave 0, one lane per 4x4 sub-block of the area */
        if (t < 64) {
            const int lg = 31 - __clz(area), S = lg == 2 ? 1 : 1 << (2 * (lg - 2));
            SvtAmdTuInfo ti = {tnz, (uint8_t)in.cand_type, (uint8_t)in.intra_luma_mode, 4 /* EB_INTRA_CHROMA_DM */, 0};
            const bool live = t < S;
            if (!live)
                ti.num_nonzero = 0;
            const uint32_t b = tnz ? coeff_bits_lanes(F.q, N, lg, ti, live, t, t & (S - 1)) : 0u;
            if (t == 0)
                F.bits = b;
        }
        __syncthreads();
        if (t == 0) {
            /* PictureFullDistortionLuma table [nz != 0][intra] + the ProductFullLoop scaling */
            const int mode = tnz == 0 ? 1 : (in.cand_type == 2 ? 2 : 0);
            unsigned long long d0 = mode == 1 ? F.pred : F.res, d1 = mode == 2 ? F.res : F.pred;
            const int dshift = size == 64 ? 4 : 2 * (7 - LG);
            d0 = (d0 + (1ull << (dshift - 1))) >> dshift;
            d1 = (d1 + (1ull << (dshift - 1))) >> dshift;
            unsigned long long tuBits = ((unsigned long long)F.bits << 10) >> 15;
            /* TuCalcCostLuma (EbRateDistortionCost.c:289) */
            const int ctx = size == N, tuIndex = size == 64 ? tu + 1 : 0;
            const unsigned long long nzRate = (tuBits << 15) + in.cbf_bits[2 + ctx], zRate = in.cbf_bits[ctx];
            const unsigned long long zCost = in.cand_type == 2 ? ~0ull : (d1 << 8) + ((((unsigned long long)in.full_lambda * zRate) + (1u << 22)) >> 23);
            const unsigned long long nzCost = (d0 << 8) + ((((unsigned long long)in.full_lambda * nzRate) + (1u << 22)) >> 23);
            const bool keep = nzCost < zCost;
            ycbf |= (uint32_t)((tnz != 0) && keep) << tuIndex;
            bits_acc += keep ? tuBits : 0;
            d0_acc += keep ? d0 : d1;
            d1_acc += d1;
            o.nz[tuIndex] = tnz;
            o.ydc[size == 64 ? tu : 0] = (int16_t)abs((int)F.q[0]);
            o.cand_nz[size == 64 ? tu : 0] = (uint16_t)tnz;
        }
        __syncthreads();
    }
    if (t == 0) {
        o.ycbf = ycbf, o.coeff_bits = bits_acc, o.dist[0] = d0_acc, o.dist[1] = d1_acc;
        out_all[blockIdx.x] = o;
    }
}

extern "C" int svt_amd_full_loop_luma_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdFullLoopIn *d_in,
                                            const int16_t *d_residual, int16_t *d_quant, int16_t *d_recon,
                                            SvtAmdFullLoopOut *d_out, uint32_t ncand)
{
    if (!ctx || !cost || !d_in || !d_residual || !d_quant || !d_recon || !d_out || !ncand)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rate_upload_tables(cost, ctx->stream);
    if (rc)
        return rc;
    /* one launch per transform size; a workgroup whose candidate has another size returns at once
     * (EstimateTransform shifts: Transform32x32Estimate 6/9 wrap 2, Transform16x16Estimate 4/9 wrap 1, Transform8x8 2/9) */
    hipLaunchKernelGGL(k_full_loop_luma<32>, dim3(ncand), dim3(TX_THREADS), 0, ctx->stream, d_in, d_residual, d_quant, d_recon,
                       d_out, 6, 9, 2);
    hipLaunchKernelGGL(k_full_loop_luma<16>, dim3(ncand), dim3(TX_THREADS), 0, ctx->stream, d_in, d_residual, d_quant, d_recon,
                       d_out, 4, 9, 1);
    hipLaunchKernelGGL(k_full_loop_luma<8>, dim3(ncand), dim3(TX_THREADS), 0, ctx->stream, d_in, d_residual, d_quant, d_recon,
                       d_out, 2, 9, 0);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* Host-pointer form for one candidate (the per-call binding of integration/svt_hook_me.c): operands are staged
 * through scratch buffers owned by the context's device; blocking.  Row pitch of the three host arrays = `pitch`
 * samples (the reference's 64-sample LCU buffers); only the (T >> pf) area of every TU is written back. */
extern "C" int svt_amd_full_loop_luma(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdFullLoopIn *in,
                                      const int16_t *residual, int16_t *quant, int16_t *recon, uint32_t pitch,
                                      SvtAmdFullLoopOut *out)
{
    if (!ctx || !cost || !in || !residual || !quant || !recon || !out || pitch < in->size ||
        (in->size != 8 && in->size != 16 && in->size != 32 && in->size != 64) || in->pf_mode > 1)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    static uint8_t *d_scratch = nullptr; /* in | out | residual | quant | recon ; callers serialise per context */
    const size_t o_in = 0, o_out = 256, o_res = 512, o_q = o_res + 8192, o_r = o_q + 8192, total = o_r + 8192;
    if (!d_scratch)
        HIP_TRY(hipMalloc((void **)&d_scratch, total));
    const uint32_t S = in->size;
    int16_t packed[64 * 64];
    for (uint32_t y = 0; y < S; y++)
        ::memcpy(packed + y * S, residual + (size_t)y * pitch, S * sizeof(int16_t));
    HIP_TRY(hipMemcpyAsync(d_scratch + o_in, in, sizeof(*in), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_scratch + o_res, packed, (size_t)S * S * 2, hipMemcpyHostToDevice, ctx->stream));
    int rc = svt_amd_full_loop_luma_batch(ctx, cost, (const SvtAmdFullLoopIn *)(d_scratch + o_in), (const int16_t *)(d_scratch + o_res),
                                          (int16_t *)(d_scratch + o_q), (int16_t *)(d_scratch + o_r),
                                          (SvtAmdFullLoopOut *)(d_scratch + o_out), 1);
    if (rc)
        return rc;
    int16_t hq[64 * 64], hr[64 * 64];
    HIP_TRY(hipMemcpyAsync(out, d_scratch + o_out, sizeof(*out), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hq, d_scratch + o_q, (size_t)S * S * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hr, d_scratch + o_r, (size_t)S * S * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const uint32_t T = S == 64 ? 32 : S, area = T >> in->pf_mode;
    for (uint32_t ty = 0; ty < S; ty += T)
        for (uint32_t tx = 0; tx < S; tx += T)
            for (uint32_t y = 0; y < area; y++) {
                ::memcpy(quant + (size_t)(ty + y) * pitch + tx, hq + (ty + y) * S + tx, area * sizeof(int16_t));
                ::memcpy(recon + (size_t)(ty + y) * pitch + tx, hr + (ty + y) * S + tx, area * sizeof(int16_t));
            }
    return SVT_AMD_OK;
}

/* ------------------------------------------------------------------------- */
/* chroma: FullLoop_R + CuFullDistortionFastTuMode_R (EbFullLoop.c:579-1066)   */
/* ------------------------------------------------------------------------- */
/* One workgroup per (candidate, plane): blockIdx.y = 0 Cb, 1 Cr.  Same on-chip chain as the luma kernel; the chroma
 * planes have no cbf cost decision (TuCalcCost, EbRateDistortionCost.c:273-279: cbf = any non-zero coefficient). */
template <int N>
__global__ __launch_bounds__(TX_THREADS) void k_full_loop_chroma(const SvtAmdChromaLoopIn *__restrict__ in_all,
                                                                const int16_t *__restrict__ residual,
                                                                int16_t *__restrict__ quant, int16_t *__restrict__ recon,
                                                                SvtAmdChromaLoopOut *__restrict__ out_all, int shift1,
                                                                int shift2, int wrap_levels)
{
    __shared__ TxShared<N> X;
    __shared__ FlShared F;
    const SvtAmdChromaLoopIn in = in_all[blockIdx.x];
    const int size = (int)in.size, T = size == 64 ? 16 : size >> 1;
    if (T != N)
        return; /* this launch serves the other transform sizes */
    const int t = threadIdx.x, plane = blockIdx.y, ntu = size == 64 ? 4 : 1, pitch = size >> 1;
    const size_t base = (size_t)blockIdx.x * 2048 + (size_t)plane * 1024;
    /* correctedPFMode (EbFullLoop.c:647-652) */
    const int pf = N == 4 ? 0 : (N == 8 && in.pf_mode == 2 ? 1 : (int)in.pf_mode);
    const int area = N >> pf;
    for (int i = t; i < 32 * 32; i += TX_THREADS)
        (&X.T[0][0])[i] = (&c_T32[0][0])[i];
    /* UnifiedQuantizeInvQuantize_R (EbFullLoop.c:483-497), bitDepth 8 */
    const uint32_t qp = plane ? in.cr_qp : in.cb_qp;
    const int qpRem = (int)(qp % 6), qpPer = (int)(qp / 6);
    const uint32_t QF = qpRem == 0 ? 26214u : qpRem == 1 ? 23302u : qpRem == 2 ? 20560u : qpRem == 3 ? 18396u : qpRem == 4 ? 16384u : 14564u;
    const int FFv = qpRem == 0 ? 40 : qpRem == 1 ? 45 : qpRem == 2 ? 51 : qpRem == 3 ? 57 : qpRem == 4 ? 64 : 72;
    constexpr int LG = N == 16 ? 4 : N == 8 ? 3 : 2;
    const int tshift = 7 - LG, shiftedQBits = 14 + qpPer + tshift;
    const uint32_t q_offset = ((in.slice_type == 2 || in.slice_type == 3) ? 171u : 85u) << (shiftedQBits - 9);
    const int shiftedFFunc = qpPer > 8 ? FFv << (qpPer - 2) : FFv << qpPer;
    const int shiftNum = qpPer > 8 ? 20 - 14 - tshift - 2 : 20 - 14 - tshift;
    const int iq_offset = 1 << (shiftNum - 1);

    uint32_t cbf = 0, nzs[5] = {0, 0, 0, 0, 0};
    unsigned long long bits_acc = 0, d0_acc = 0, d1_acc = 0;

    for (int tu = 0; tu < ntu; tu++) {
        const int off = ntu == 4 ? ((tu & 1) << 4) + (tu > 1 ? 16 * 32 : 0) : 0;
        constexpr int GB = TxShared<N>::GB;
        for (int i = t; i < GB * N * N; i += TX_THREADS)
            (&X.io[0][0])[i] = i < N * N ? residual[base + off + (i / N) * pitch + (i % N)] : (int16_t)0;
        if (t == 0)
            F.nz = 0, F.res = 0, F.pred = 0, F.bits = 0;
        __syncthreads();
        fwd_pass<N, false>(X, shift1, wrap_levels, nullptr, 1, t);
        fwd_pass<N, false>(X, shift2, wrap_levels, nullptr, 1, t);
        unsigned nz = 0, res = 0, pred = 0;
        for (int i = t; i < area * area; i += TX_THREADS) {
            const int r = i / area, c = i - r * area;
            const int v = X.io[0][r * N + c], sign = v < 0 ? -1 : 1;
            int tq = abs(v);
            tq = (int)((uint32_t)tq * QF);
            tq = (int)((uint32_t)tq + q_offset);
            tq >>= shiftedQBits;
            const int qv = clip16i(sign * tq);
            const int rv = clip16i(((qv * shiftedFFunc) + iq_offset) >> shiftNum);
            F.q[r * N + c] = (int16_t)qv;
            quant[base + off + r * pitch + c] = (int16_t)qv;
            recon[base + off + r * pitch + c] = (int16_t)rv;
            nz += qv != 0;
            const int16_t d = (int16_t)(v - rv);
            res += (unsigned)(d * d);
            pred += (unsigned)(v * v);
        }
        for (int s = 32; s > 0; s >>= 1)
            nz += __shfl_xor(nz, s), res += __shfl_xor(res, s), pred += __shfl_xor(pred, s);
        if ((t & 63) == 0) {
            atomicAdd(&F.nz, nz);
            atomicAdd(&F.res, res);
            atomicAdd(&F.pred, pred);
        }
        __syncthreads();
        const unsigned tnz = F.nz;
        /* TuEstimateCoeffBits_R, chroma branch: wave 0, one lane per 4x4 sub-block of the area */
        if (t < 64) {
            const int lg = 31 - __clz(area), S = lg == 2 ? 1 : 1 << (2 * (lg - 2));
            SvtAmdTuInfo ti = {tnz, (uint8_t)in.cand_type, (uint8_t)in.intra_luma_mode, 4 /* EB_INTRA_CHROMA_DM */,
                               (uint8_t)(plane + 1)};
            const bool live = t < S;
            if (!live)
                ti.num_nonzero = 0;
            const uint32_t b = tnz ? coeff_bits_lanes(F.q, N, lg, ti, live, t, t & (S - 1)) : 0u;
            if (t == 0)
                F.bits = b;
        }
        __syncthreads();
        if (t == 0) {
            /* PictureFullDistortion_R table [nz != 0][intra] + the chroma scaling (EbFullLoop.c:1000-1004) */
            const int mode = tnz == 0 ? 1 : (in.cand_type == 2 ? 2 : 0);
            unsigned long long d0 = mode == 1 ? F.pred : F.res, d1 = mode == 2 ? F.res : F.pred;
            const int dshift = 2 * (7 - LG);
            d0_acc += (d0 + (1ull << (dshift - 1))) >> dshift;
            d1_acc += (d1 + (1ull << (dshift - 1))) >> dshift;
            bits_acc += ((unsigned long long)F.bits << 10) >> 15;
            const int tuIndex = ntu == 4 ? tu + 1 : 0;
            cbf |= (uint32_t)(tnz != 0) << tuIndex;
            nzs[tuIndex] = tnz;
        }
        __syncthreads();
    }
    if (t == 0) {
        SvtAmdChromaLoopOut *o = out_all + blockIdx.x;
        for (int k = 0; k < 5; k++)
            o->nz[plane][k] = nzs[k];
        o->cbf[plane] = cbf, o->coeff_bits[plane] = bits_acc, o->dist[plane][0] = d0_acc, o->dist[plane][1] = d1_acc;
    }
}

extern "C" int svt_amd_full_loop_chroma_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdChromaLoopIn *d_in,
                                              const int16_t *d_residual, int16_t *d_quant, int16_t *d_recon,
                                              SvtAmdChromaLoopOut *d_out, uint32_t ncand)
{
    if (!ctx || !cost || !d_in || !d_residual || !d_quant || !d_recon || !d_out || !ncand)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rate_upload_tables(cost, ctx->stream);
    if (rc)
        return rc;
    /* EstimateTransform shifts: Transform16x16Estimate 4/9 wrap 1, Transform8x8 2/9, Transform4x4 1/8 */
    hipLaunchKernelGGL(k_full_loop_chroma<16>, dim3(ncand, 2), dim3(TX_THREADS), 0, ctx->stream, d_in, d_residual, d_quant,
                       d_recon, d_out, 4, 9, 1);
    hipLaunchKernelGGL(k_full_loop_chroma<8>, dim3(ncand, 2), dim3(TX_THREADS), 0, ctx->stream, d_in, d_residual, d_quant,
                       d_recon, d_out, 2, 9, 0);
    hipLaunchKernelGGL(k_full_loop_chroma<4>, dim3(ncand, 2), dim3(TX_THREADS), 0, ctx->stream, d_in, d_residual, d_quant,
                       d_recon, d_out, 1, 8, 0);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

extern "C" int svt_amd_full_loop_chroma(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdChromaLoopIn *in,
                                        const int16_t *const residual[2], int16_t *const quant[2], int16_t *const recon[2],
                                        uint32_t pitch, SvtAmdChromaLoopOut *out)
{
    if (!ctx || !cost || !in || !residual || !quant || !recon || !out || !residual[0] || !residual[1] || !quant[0] ||
        !quant[1] || !recon[0] || !recon[1] || pitch < in->size / 2 ||
        (in->size != 8 && in->size != 16 && in->size != 32 && in->size != 64) || in->pf_mode > 2)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    static uint8_t *d_scratch = nullptr; /* in | out | residual | quant | recon ; callers serialise per context */
    const size_t o_in = 0, o_out = 256, o_res = 512, o_q = o_res + 4096, o_r = o_q + 4096, total = o_r + 4096;
    if (!d_scratch)
        HIP_TRY(hipMalloc((void **)&d_scratch, total));
    const uint32_t C = in->size / 2;
    int16_t packed[2048];
    for (int p = 0; p < 2; p++)
        for (uint32_t y = 0; y < C; y++)
            ::memcpy(packed + p * 1024 + y * C, residual[p] + (size_t)y * pitch, C * sizeof(int16_t));
    HIP_TRY(hipMemcpyAsync(d_scratch + o_in, in, sizeof(*in), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_scratch + o_res, packed, sizeof(packed), hipMemcpyHostToDevice, ctx->stream));
    int rc = svt_amd_full_loop_chroma_batch(ctx, cost, (const SvtAmdChromaLoopIn *)(d_scratch + o_in),
                                            (const int16_t *)(d_scratch + o_res), (int16_t *)(d_scratch + o_q),
                                            (int16_t *)(d_scratch + o_r), (SvtAmdChromaLoopOut *)(d_scratch + o_out), 1);
    if (rc)
        return rc;
    int16_t hq[2048], hr[2048];
    HIP_TRY(hipMemcpyAsync(out, d_scratch + o_out, sizeof(*out), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hq, d_scratch + o_q, sizeof(hq), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hr, d_scratch + o_r, sizeof(hr), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const uint32_t T = in->size == 64 ? 16 : C;
    const uint32_t area = T >> (T == 4 ? 0 : (T == 8 && in->pf_mode == 2 ? 1 : in->pf_mode));
    for (int p = 0; p < 2; p++)
        for (uint32_t ty = 0; ty < C; ty += T)
            for (uint32_t tx = 0; tx < C; tx += T)
                for (uint32_t y = 0; y < area; y++) {
                    ::memcpy(quant[p] + (size_t)(ty + y) * pitch + tx, hq + p * 1024 + (ty + y) * C + tx, area * sizeof(int16_t));
                    ::memcpy(recon[p] + (size_t)(ty + y) * pitch + tx, hr + p * 1024 + (ty + y) * C + tx, area * sizeof(int16_t));
                }
    return SVT_AMD_OK;
}
