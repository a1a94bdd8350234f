/*
 * Luma and chroma full loops of mode-decision candidates, fused (EncDec row "PerformFullLoop, ProductFullLoop ...").
 *
 * Replaces ProductFullLoop (Codec/EbFullLoop.c:185-446) and the pair FullLoop_R + CuFullDistortionFastTuMode_R (:579-1066)
 * for the presets' common configuration (plain quantiser or, for the luma loop, its PM-core variant of encMode 1..4; no RDOQ;
 * coefficient-domain distortion, no CABAC-context update).
 * One wave owns 64 / N transform units at a time, each on N lanes of its own: lane r holds row r of the residual through the
 * first transform pass and column r of the coefficient block from the second pass on (txfm_device.h, register-resident
 * transform: the matrix is immediates, LDS only carries the transpose); a 64x64 CU walks its four units one after the
 * other.  Everything between the residual and the cost decision
 * stays on chip:
 *   residual -> LDS -> forward "Estimate" DCT (two passes in LDS, txfm_device.h)
 *            -> quantise / inverse-quantise the (T >> pf) area, count non-zeros, coefficient-domain distortions
 *            -> coefficient bits (one lane per 4x4 sub-block, rate_device.h) -> cbf / cost decision (one thread).
 * Only the quantised and reconstructed coefficients and a 64-byte result record leave the workgroup.
 * HBM traffic per TU: 2 B/sample in, 4 B/sample out (area only) - the five-kernel composition moves 14 B/sample.
 */
#include "txfm_device.h"
#include "rate_device.h"
#include "pmcore_device.h"
#include <cstring>

/* N: transform size of this launch.  CHROMA 0: SvtAmdFullLoopIn/Out, residual slab 4096 samples per candidate, one unit
 * sequence per candidate; CHROMA 1: SvtAmdChromaLoopIn/Out, slab 2048 (Cb then Cr), one unit sequence per (candidate,
 * plane).  One wave per workgroup; the wave's 64 / N unit sequences run side by side, each on its own N lanes; a
 * 64x64 CU's sequence has four units (all others one).  No workgroup barrier anywhere: a unit never leaves its wave. */
template <int N, bool CHROMA, bool PM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PM ? 2 : 1))) void k_full_loop(const SvtAmdCabacCost *__restrict__ cost_p, const void *__restrict__ in_all, const int16_t *__restrict__ residual,
                                                  int16_t *__restrict__ quant, int16_t *__restrict__ recon,
                                                  void *__restrict__ out_all, uint32_t ncand, int shift1, int shift2,
                                                  int wrap_levels, uint32_t *__restrict__ ctx_models)
{
    const SvtAmdCabacCost &c_cost = *cost_p;
    constexpr int UPW = 64 / N; /* unit sequences per wave */
    constexpr int LG = N == 32 ? 5 : N == 16 ? 4 : N == 8 ? 3 : 2;
    __shared__ int16_t tiles[UPW * TxRegTile<N>::UNIT];
    __shared__ int16_t Fq[UPW][N * N]; /* quantised coefficients of the units in flight, row pitch N */
    __shared__ uint32_t Fbits[UPW];
    __shared__ int16_t Pq[PM ? 64 : 1][16]; /* PM-core: the 4x4 block of levels a lane is pricing */
    /* coeffCabacUpdate (ctx_models != NULL): the candidate's CoeffCtxtMdl_t, threaded through its units in the reference's order
     * (luma: unit after unit; chroma: Cb then Cr of every unit - the two planes of a candidate are neighbouring sequences) */
    __shared__ uint8_t Mctx[UPW][RATE_CTX_WORDS];
    __shared__ uint16_t Msig[UPW][64], MabsC[UPW][16];
    const int t = threadIdx.x, u = t / N, r = t - u * N;
    const int mu = CHROMA ? (u & ~1) : u; /* the sequence whose Mctx slot holds this candidate's model */
    const uint32_t seq = blockIdx.x * UPW + u;
    const uint32_t cand = CHROMA ? seq >> 1 : seq, plane = CHROMA ? seq & 1 : 0;

    int size = 0, ntu = 0, cand_type = 0, intra_luma_mode = 0, pm_core = 0;
    uint32_t pf = 0, qp = 0, slice = 0;
    if (cand < ncand) {
        if (CHROMA) {
            const SvtAmdChromaLoopIn *in = (const SvtAmdChromaLoopIn *)in_all + cand;
            size = (int)in->size;
            if ((size == 64 ? 16 : size >> 1) == N) {
                ntu = size == 64 ? 4 : 1;
                /* correctedPFMode (EbFullLoop.c:647-652): 4x4 never, 8x8 at most N2 */
                pf = N == 4 ? 0u : (N == 8 && in->pf_mode == 2 ? 1u : in->pf_mode);
                qp = plane ? in->cr_qp : in->cb_qp, slice = in->slice_type;
                cand_type = (int)in->cand_type, intra_luma_mode = (int)in->intra_luma_mode;
            }
        } else {
            const SvtAmdFullLoopIn *in = (const SvtAmdFullLoopIn *)in_all + cand;
            size = (int)in->size;
            if ((size == 64 ? 32 : size) == N && (in->pm_core != 0) == PM) { /* PM-core candidates have their own launch */
                ntu = size == 64 ? 4 : 1;
                pf = in->pf_mode, qp = in->qp, slice = in->slice_type, pm_core = in->pm_core;
                cand_type = (int)in->cand_type, intra_luma_mode = (int)in->intra_luma_mode;
            }
        }
    }
    /* this launch serves one transform size: a wave none of whose candidates has it leaves at once */
    if (!__ballot(ntu != 0))
        return;
    int ntu_max = ntu;
    for (int o = 32; o > 0; o >>= 1)
        ntu_max = max(ntu_max, __shfl_xor(ntu_max, o));

    if (ctx_models && ntu && (!CHROMA || plane == 0))
        for (int i = r; i < RATE_CTX_WORDS; i += N)
            Mctx[u][i] = (uint8_t)ctx_models[(size_t)cand * RATE_CTX_WORDS + i];
    /* running results of the sequence (meaningful in lane 0 of the unit) */
    uint32_t cbf = 0, nzs[5] = {0, 0, 0, 0, 0};
    unsigned long long bits_acc = 0, d0_acc = 0, d1_acc = 0;
    int16_t ydc[4] = {0, 0, 0, 0};
    uint16_t cand_nz[4] = {0, 0, 0, 0};
    uint32_t cbf_bits0 = 0, cbf_bits1 = 0, full_lambda = 0;
    if (!CHROMA && ntu) {
        const SvtAmdFullLoopIn *in = (const SvtAmdFullLoopIn *)in_all + cand;
        cbf = in->ycbf, bits_acc = in->coeff_bits;
        d0_acc = size == 64 ? in->dist[0] : 0, d1_acc = size == 64 ? in->dist[1] : 0;
        const int ctx = size == N;
        cbf_bits0 = in->cbf_bits[ctx], cbf_bits1 = in->cbf_bits[2 + ctx], full_lambda = in->full_lambda;
    }

    FlUnit S;
    S.area = N >> pf, S.lg = LG - (int)pf;
    S.pitch = CHROMA ? (uint32_t)size >> 1 : (uint32_t)size;
    fl_quant_params(S, qp, slice, LG);
    int16_t *tile = tiles + u * TxRegTile<N>::UNIT;

    for (int tu = 0; tu < ntu_max; tu++) {
        S.active = tu < ntu;
        if (CHROMA)
            S.base = cand * 2048u + plane * 1024u + (size == 64 ? ((tu & 1) << 4) + (tu > 1 ? 16 * 32 : 0) : 0);
        else
            S.base = cand * 4096u + (size == 64 ? ((tu & 1) << 5) + (tu > 1 ? 32 * 64 : 0) : 0);
        /* residual row r -> registers */
        int x[N];
        if (S.active) {
            const int16_t *row = residual + S.base + (size_t)r * S.pitch;
            if (N >= 8) {
#pragma unroll
                for (int j = 0; j < N; j += 8) {
                    const uint4 v = *(const uint4 *)(row + j);
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        x[j + 2 * k] = (int16_t)(w[k] & 0xffffu), x[j + 2 * k + 1] = (int16_t)(w[k] >> 16);
                }
            } else {
                const uint2 v = *(const uint2 *)row;
                x[0] = (int16_t)(v.x & 0xffffu), x[1] = (int16_t)(v.x >> 16), x[2] = (int16_t)(v.y & 0xffffu), x[3] = (int16_t)(v.y >> 16);
            }
        } else {
#pragma unroll
            for (int j = 0; j < N; j++)
                x[j] = 0;
        }
        /* EstimateTransform: both passes in registers; afterwards x[j] = coefficient (j, r) */
        fwd_2d_regs<N>(x, tile, r, shift1, shift2, wrap_levels);
        /* QuantizeInvQuantize (C_DEFAULT/EbTransforms_C.c:89) over the area + the two coefficient-domain sums */
        unsigned nz = 0, res = 0, pred = 0;
        if (S.active && r < S.area) {
#pragma unroll
            for (int j = 0; j < N; j++) {
                if (j < S.area) {
                    const int v = x[j], sign = v < 0 ? -1 : 1;
                    int tq = abs(v);
                    tq = (int)((uint32_t)tq * S.QF);
                    tq = (int)((uint32_t)tq + S.q_offset);
                    tq >>= S.shiftedQBits;
                    const int qv = clip16i(sign * tq);
                    const int rv = clip16i(((qv * S.shiftedFFunc) + S.iq_offset) >> S.shiftNum);
                    Fq[u][j * N + r] = (int16_t)qv;
                    quant[S.base + j * S.pitch + r] = (int16_t)qv;
                    recon[S.base + j * S.pitch + r] = (int16_t)rv;
                    const int16_t d = (int16_t)(v - rv);
                    nz += qv != 0, res += (unsigned)(d * d), pred += (unsigned)(v * v);
                }
            }
        }
#pragma unroll
        for (int o = 1; o < N; o <<= 1)
            nz += __shfl_xor(nz, o), res += __shfl_xor(res, o), pred += __shfl_xor(pred, o);
        if constexpr (PM) {
            /* DecoupledQuantizeInvQuantizeLoops, EB_PMCORE branch (Codec/EbTransforms.c:2807-2950): every 4x4 block of the area
             * that holds a level is re-quantised from its coefficients scaled by 100 / 70 / 50 % (MatMultOut :39-71; the DC of
             * block 0 passes unscaled when its regular level exceeds PM_DC_TRSHLD1) and the cheapest of the three in
             * coefficient-domain SSE + lambda * (4x4 rate estimate) replaces it.  One lane per 4x4 block. */
            const bool pmu = pm_core != 0 && S.active && nz != 0;
            if (__ballot(pmu)) {
                if (pmu && r < S.area) {
#pragma unroll
                    for (int j = 0; j < N; j++)
                        if (j < S.area)
                            tile[j * N + r] = (int16_t)x[j];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                pm_core_blocks<N>(c_cost, tile, &Fq[u][0], N, S.area, S.lg, r, pmu, t, cand_type, full_lambda, S, Pq);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                /* the levels are final now: de-quantise again, rewrite the outputs, recount */
                unsigned nz2 = 0, res2 = 0, pred2 = 0;
                if (pmu && r < S.area) {
#pragma unroll
                    for (int j = 0; j < N; j++) {
                        if (j < S.area) {
                            /* the coefficient comes back from the tile: x[] need not stay in registers across the re-decision */
                            const int v = (int)tile[j * N + r], qv = (int)Fq[u][j * N + r];
                            const int rv = clip16i(((qv * S.shiftedFFunc) + S.iq_offset) >> S.shiftNum);
                            quant[S.base + j * S.pitch + r] = (int16_t)qv;
                            recon[S.base + j * S.pitch + r] = (int16_t)rv;
                            const int16_t d = (int16_t)(v - rv);
                            nz2 += qv != 0, res2 += (unsigned)(d * d), pred2 += (unsigned)(v * v);
                        }
                    }
                }
#pragma unroll
                for (int o = 1; o < N; o <<= 1)
                    nz2 += __shfl_xor(nz2, o), res2 += __shfl_xor(res2, o), pred2 += __shfl_xor(pred2, o);
                if (pmu)
                    nz = nz2, res = res2, pred = pred2;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (ctx_models) {
            /* context-updating estimator: the unit's lanes build its significance maps, lane 0 walks (rate_device.h) */
            const SvtAmdTuInfo ti = {nz, (uint8_t)cand_type, (uint8_t)intra_luma_mode, 4 /* EB_INTRA_CHROMA_DM */,
                                     CHROMA ? (uint8_t)(plane + 1) : (uint8_t)0};
            if (S.active && nz != 0)
                rate_update_sigmaps(Msig[u], &Fq[u][0], N, S.lg, ti, r, N);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
            for (uint32_t ph = 0; ph < (CHROMA ? 2u : 1u); ph++) {
                if (r == 0 && S.active && (!CHROMA || plane == ph))
                    Fbits[u] = nz ? coeff_bits_update_walk(Mctx[mu], Msig[u], MabsC[u], &Fq[u][0], N, S.lg, ti) : 0u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        } else {
            /* TuEstimateCoeffBitsLuma / TuEstimateCoeffBits_R: one lane per 4x4 sub-block; units of equal area together,
             * as many per call as the wave holds */
            if (r == 0)
                Fbits[u] = 0;
    #pragma unroll 1
            for (int pfv = 0; pfv < (N == 4 ? 1 : 3); pfv++) {
                const int lg = LG - pfv;
                if (lg < 2)
                    break;
                if (!__ballot(S.active && S.lg == lg && nz != 0))
                    continue;
                const int S4 = lg == 2 ? 1 : 1 << (2 * (lg - 2));
                const int per_call = 64 / S4 < UPW ? 64 / S4 : UPW;
    #pragma unroll 1
                for (int c0 = 0; c0 < UPW; c0 += per_call) {
                    const int uu = c0 + t / S4, sub = t % S4;
                    const int src_lane = (uu < UPW ? uu : 0) * N; /* lane 0 of that unit holds its facts */
                    const int u_ok = __shfl(S.active && S.lg == lg && nz != 0, src_lane);
                    const uint32_t u_nz = (uint32_t)__shfl((int)nz, src_lane);
                    const int u_type = __shfl(cand_type, src_lane), u_mode = __shfl(intra_luma_mode, src_lane);
                    const int u_plane = __shfl((int)plane, src_lane);
                    const bool live = uu < c0 + per_call && uu < UPW && u_ok;
                    if (!__ballot(live))
                        continue;
                    SvtAmdTuInfo ti = {live ? u_nz : 0u, (uint8_t)u_type, (uint8_t)u_mode, 4 /* EB_INTRA_CHROMA_DM */,
                                       CHROMA ? (uint8_t)(u_plane + 1) : (uint8_t)0};
                    const uint32_t b = coeff_bits_lanes(c_cost, &Fq[uu < UPW ? uu : 0][0], N, lg, ti, live, t, sub);
                    if (live && sub == 0)
                        Fbits[uu] = b;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (r == 0 && S.active) {
            /* PictureFullDistortionLuma / _R table [nz != 0][intra] + the scaling of the caller */
            const int mode = nz == 0 ? 1 : (cand_type == 2 ? 2 : 0);
            unsigned long long d0 = mode == 1 ? pred : res, d1 = mode == 2 ? res : pred;
            const int dshift = (!CHROMA && size == 64) ? 4 : 2 * (7 - LG);
            d0 = (d0 + (1ull << (dshift - 1))) >> dshift;
            d1 = (d1 + (1ull << (dshift - 1))) >> dshift;
            const unsigned long long tuBits = ctx_models ? (unsigned long long)(Fbits[u] >> 15) : ((unsigned long long)Fbits[u] << 10) >> 15;
            const int tuIndex = size == 64 ? tu + 1 : 0;
            nzs[tuIndex] = nz;
            if (CHROMA) {
                /* TuCalcCost, chroma branches (EbRateDistortionCost.c:273-279) */
                cbf |= (uint32_t)(nz != 0) << tuIndex;
                bits_acc += tuBits, d0_acc += d0, d1_acc += d1;
            } else {
                /* TuCalcCostLuma (EbRateDistortionCost.c:289) */
                const unsigned long long nzRate = (tuBits << 15) + cbf_bits1, zRate = cbf_bits0, lam = full_lambda;
                const unsigned long long zCost = cand_type == 2 ? ~0ull : (d1 << 8) + (((lam * zRate) + (1u << 22)) >> 23);
                const unsigned long long nzCost = (d0 << 8) + (((lam * nzRate) + (1u << 22)) >> 23);
                const bool keep = nzCost < zCost;
                cbf |= (uint32_t)((nz != 0) && keep) << tuIndex;
                bits_acc += keep ? tuBits : 0;
                d0_acc += keep ? d0 : d1;
                d1_acc += d1;
                ydc[size == 64 ? tu : 0] = (int16_t)abs((int)Fq[u][0]);
                cand_nz[size == 64 ? tu : 0] = (uint16_t)nz;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (ctx_models && ntu && (!CHROMA || plane == 0)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = r; i < RATE_CTX_WORDS; i += N)
            ctx_models[(size_t)cand * RATE_CTX_WORDS + i] = Mctx[u][i];
    }
    if (r == 0 && ntu) {
        if (CHROMA) {
            SvtAmdChromaLoopOut *o = (SvtAmdChromaLoopOut *)out_all + cand;
            for (int k = 0; k < 5; k++)
                o->nz[plane][k] = nzs[k];
            o->cbf[plane] = cbf, o->coeff_bits[plane] = bits_acc, o->dist[plane][0] = d0_acc, o->dist[plane][1] = d1_acc;
        } else {
            SvtAmdFullLoopOut o;
            for (int k = 0; k < 5; k++)
                o.nz[k] = nzs[k];
            for (int k = 0; k < 4; k++)
                o.ydc[k] = ydc[k], o.cand_nz[k] = cand_nz[k];
            o.ycbf = cbf, o.coeff_bits = bits_acc, o.dist[0] = d0_acc, o.dist[1] = d1_acc;
            ((SvtAmdFullLoopOut *)out_all)[cand] = o;
        }
    }
}

template <int N, bool CHROMA, bool PM = false>
static void launch_full_loop(SvtAmdContext *ctx, const void *d_in, const int16_t *d_res, int16_t *d_q, int16_t *d_r, void *d_out,
                             uint32_t ncand, int s1, int s2, int wrap, uint32_t *d_models = nullptr)
{
    constexpr int UPW = 64 / N;
    const uint32_t nseq = CHROMA ? 2 * ncand : ncand;
    hipLaunchKernelGGL((k_full_loop<N, CHROMA, PM>), dim3((nseq + UPW - 1) / UPW), dim3(64), 0, ctx->stream, (const SvtAmdCabacCost *)ctx->d_cabac_cost, d_in, d_res, d_q, d_r, d_out, ncand,
                       s1, s2, wrap, d_models);
}

extern "C" int svt_amd_full_loop_luma_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdFullLoopIn *d_in,
                                            const int16_t *d_residual, int16_t *d_quant, int16_t *d_recon,
                                            SvtAmdFullLoopOut *d_out, uint32_t ncand)
{
    if (!ctx || !cost || !d_in || !d_residual || !d_quant || !d_recon || !d_out || !ncand)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rate_upload_tables(ctx, cost);
    if (rc)
        return rc;
    /* one launch per transform size; a workgroup none of whose candidates has that size returns at once (sorting the
     * batch by CU size keeps the 8x8 workgroups full).
     * EstimateTransform shifts: Transform32x32Estimate 6/9 wrap 2, Transform16x16Estimate 4/9 wrap 1, Transform8x8 2/9 */
    launch_full_loop<32, false>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 6, 9, 2);
    launch_full_loop<16, false>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 4, 9, 1);
    launch_full_loop<8, false>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 2, 9, 0);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* The candidates of the batch whose pm_core is set (pictures of encMode 1..4: contextPtr->rdoqPmCoreMethod == EB_PMCORE is a
 * picture-level switch, EbEncDecProcess.c:2201); the plain entry point above serves the others.  Kept apart because the block
 * re-decision costs registers the plain path should not pay for. */
extern "C" int svt_amd_full_loop_luma_pmcore_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdFullLoopIn *d_in,
                                                   const int16_t *d_residual, int16_t *d_quant, int16_t *d_recon,
                                                   SvtAmdFullLoopOut *d_out, uint32_t ncand)
{
    if (!ctx || !cost || !d_in || !d_residual || !d_quant || !d_recon || !d_out || !ncand)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rate_upload_tables(ctx, cost);
    if (rc)
        return rc;
    launch_full_loop<32, false, true>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 6, 9, 2);
    launch_full_loop<16, false, true>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 4, 9, 1);
    launch_full_loop<8, false, true>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 2, 9, 0);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

extern "C" int svt_amd_full_loop_chroma_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdChromaLoopIn *d_in,
                                              const int16_t *d_residual, int16_t *d_quant, int16_t *d_recon,
                                              SvtAmdChromaLoopOut *d_out, uint32_t ncand)
{
    if (!ctx || !cost || !d_in || !d_residual || !d_quant || !d_recon || !d_out || !ncand)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rate_upload_tables(ctx, cost);
    if (rc)
        return rc;
    /* EstimateTransform shifts: Transform16x16Estimate 4/9 wrap 1, Transform8x8 2/9, Transform4x4 1/8 */
    launch_full_loop<16, true>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 4, 9, 1);
    launch_full_loop<8, true>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 2, 9, 0);
    launch_full_loop<4, true>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 1, 8, 0);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* coeffCabacUpdate forms (EbEncDecProcess.c:2115-2123; EbFullLoop.c:265-280, 417-432, 1014-1035): the units' coefficient bits come
 * from the context-updating estimator and every candidate's CoeffCtxtMdl_t (d_ctx_models[cand], SVT_AMD_COEFF_CTX_WORDS words,
 * = candidateBuffer->candBuffCoeffCtxModel) is updated in place.  Plain and PM-core candidates may be mixed. */
extern "C" int svt_amd_full_loop_luma_cabac_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdFullLoopIn *d_in,
                                                  const int16_t *d_residual, int16_t *d_quant, int16_t *d_recon,
                                                  SvtAmdFullLoopOut *d_out, uint32_t *d_ctx_models, uint32_t ncand)
{
    if (!ctx || !cost || !d_in || !d_residual || !d_quant || !d_recon || !d_out || !d_ctx_models || !ncand)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rate_upload_tables(ctx, cost);
    if (rc)
        return rc;
    launch_full_loop<32, false>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 6, 9, 2, d_ctx_models);
    launch_full_loop<16, false>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 4, 9, 1, d_ctx_models);
    launch_full_loop<8, false>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 2, 9, 0, d_ctx_models);
    launch_full_loop<32, false, true>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 6, 9, 2, d_ctx_models);
    launch_full_loop<16, false, true>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 4, 9, 1, d_ctx_models);
    launch_full_loop<8, false, true>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 2, 9, 0, d_ctx_models);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

extern "C" int svt_amd_full_loop_chroma_cabac_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdChromaLoopIn *d_in,
                                                    const int16_t *d_residual, int16_t *d_quant, int16_t *d_recon,
                                                    SvtAmdChromaLoopOut *d_out, uint32_t *d_ctx_models, uint32_t ncand)
{
    if (!ctx || !cost || !d_in || !d_residual || !d_quant || !d_recon || !d_out || !d_ctx_models || !ncand)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rate_upload_tables(ctx, cost);
    if (rc)
        return rc;
    launch_full_loop<16, true>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 4, 9, 1, d_ctx_models);
    launch_full_loop<8, true>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 2, 9, 0, d_ctx_models);
    launch_full_loop<4, true>(ctx, d_in, d_residual, d_quant, d_recon, d_out, ncand, 1, 8, 0, d_ctx_models);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* Host-pointer form for one candidate (the per-call binding of integration/svt_hook_me.c): operands are staged
 * through scratch buffers owned by the context's device; blocking.  Row pitch of the three host arrays = `pitch`
 * samples (the reference's 64-sample LCU buffers); only the (T >> pf) area of every TU is written back. */
static int full_loop_luma_host(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdFullLoopIn *in,
                               const int16_t *residual, int16_t *quant, int16_t *recon, uint32_t pitch, uint32_t *model,
                               SvtAmdFullLoopOut *out)
{
    if (!ctx || !cost || !in || !residual || !quant || !recon || !out || pitch < in->size ||
        (in->size != 8 && in->size != 16 && in->size != 32 && in->size != 64) || in->pf_mode > 1 || (in->pm_core != 0 && in->pm_core != 2))
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *d_scratch = nullptr; /* in | out | residual | quant | recon ; callers serialise per context */
    const size_t o_in = 0, o_out = 256, o_res = 512, o_q = o_res + 8192, o_r = o_q + 8192, o_m = o_r + 8192, total = o_m + 1024;
    {
        int rc_s = svt_amd_ctx_scratch(ctx, total, &d_scratch);
        if (rc_s)
            return rc_s;
    }
    const uint32_t S = in->size;
    int16_t packed[64 * 64];
    for (uint32_t y = 0; y < S; y++)
        ::memcpy(packed + y * S, residual + (size_t)y * pitch, S * sizeof(int16_t));
    HIP_TRY(hipMemcpyAsync(d_scratch + o_in, in, sizeof(*in), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_scratch + o_res, packed, (size_t)S * S * 2, hipMemcpyHostToDevice, ctx->stream));
    int rc;
    if (model) {
        HIP_TRY(hipMemcpyAsync(d_scratch + o_m, model, RATE_CTX_WORDS * 4, hipMemcpyHostToDevice, ctx->stream));
        rc = svt_amd_full_loop_luma_cabac_batch(ctx, cost, (const SvtAmdFullLoopIn *)(d_scratch + o_in), (const int16_t *)(d_scratch + o_res),
                                                (int16_t *)(d_scratch + o_q), (int16_t *)(d_scratch + o_r),
                                                (SvtAmdFullLoopOut *)(d_scratch + o_out), (uint32_t *)(d_scratch + o_m), 1);
    } else
        rc = (in->pm_core ? svt_amd_full_loop_luma_pmcore_batch : svt_amd_full_loop_luma_batch)(
            ctx, cost, (const SvtAmdFullLoopIn *)(d_scratch + o_in), (const int16_t *)(d_scratch + o_res), (int16_t *)(d_scratch + o_q),
            (int16_t *)(d_scratch + o_r), (SvtAmdFullLoopOut *)(d_scratch + o_out), 1);
    if (rc)
        return rc;
    if (model)
        HIP_TRY(hipMemcpyAsync(model, d_scratch + o_m, RATE_CTX_WORDS * 4, hipMemcpyDeviceToHost, ctx->stream));
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

extern "C" int svt_amd_full_loop_luma(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdFullLoopIn *in,
                                      const int16_t *residual, int16_t *quant, int16_t *recon, uint32_t pitch,
                                      SvtAmdFullLoopOut *out)
{
    return full_loop_luma_host(ctx, cost, in, residual, quant, recon, pitch, nullptr, out);
}
/* coeffCabacUpdate: ctx_model = the candidate's CoeffCtxtMdl_t (SVT_AMD_COEFF_CTX_WORDS words), updated in place */
extern "C" int svt_amd_full_loop_luma_cabac(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdFullLoopIn *in,
                                            const int16_t *residual, int16_t *quant, int16_t *recon, uint32_t pitch,
                                            uint32_t *ctx_model, SvtAmdFullLoopOut *out)
{
    if (!ctx_model)
        return SVT_AMD_ERR_BAD_PARAM;
    return full_loop_luma_host(ctx, cost, in, residual, quant, recon, pitch, ctx_model, out);
}

static int full_loop_chroma_host(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdChromaLoopIn *in,
                                 const int16_t *const residual[2], int16_t *const quant[2], int16_t *const recon[2],
                                 uint32_t pitch, uint32_t *model, SvtAmdChromaLoopOut *out)
{
    if (!ctx || !cost || !in || !residual || !quant || !recon || !out || !residual[0] || !residual[1] || !quant[0] ||
        !quant[1] || !recon[0] || !recon[1] || pitch < in->size / 2 ||
        (in->size != 8 && in->size != 16 && in->size != 32 && in->size != 64) || in->pf_mode > 2)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *d_scratch = nullptr; /* in | out | residual | quant | recon ; callers serialise per context */
    const size_t o_in = 0, o_out = 256, o_res = 512, o_q = o_res + 4096, o_r = o_q + 4096, o_m = o_r + 4096, total = o_m + 1024;
    {
        int rc_s = svt_amd_ctx_scratch(ctx, total, &d_scratch);
        if (rc_s)
            return rc_s;
    }
    const uint32_t C = in->size / 2;
    int16_t packed[2048];
    for (int p = 0; p < 2; p++)
        for (uint32_t y = 0; y < C; y++)
            ::memcpy(packed + p * 1024 + y * C, residual[p] + (size_t)y * pitch, C * sizeof(int16_t));
    HIP_TRY(hipMemcpyAsync(d_scratch + o_in, in, sizeof(*in), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_scratch + o_res, packed, sizeof(packed), hipMemcpyHostToDevice, ctx->stream));
    int rc;
    if (model) {
        HIP_TRY(hipMemcpyAsync(d_scratch + o_m, model, RATE_CTX_WORDS * 4, hipMemcpyHostToDevice, ctx->stream));
        rc = svt_amd_full_loop_chroma_cabac_batch(ctx, cost, (const SvtAmdChromaLoopIn *)(d_scratch + o_in),
                                                  (const int16_t *)(d_scratch + o_res), (int16_t *)(d_scratch + o_q),
                                                  (int16_t *)(d_scratch + o_r), (SvtAmdChromaLoopOut *)(d_scratch + o_out),
                                                  (uint32_t *)(d_scratch + o_m), 1);
    } else
        rc = svt_amd_full_loop_chroma_batch(ctx, cost, (const SvtAmdChromaLoopIn *)(d_scratch + o_in),
                                            (const int16_t *)(d_scratch + o_res), (int16_t *)(d_scratch + o_q),
                                            (int16_t *)(d_scratch + o_r), (SvtAmdChromaLoopOut *)(d_scratch + o_out), 1);
    if (rc)
        return rc;
    if (model)
        HIP_TRY(hipMemcpyAsync(model, d_scratch + o_m, RATE_CTX_WORDS * 4, hipMemcpyDeviceToHost, ctx->stream));
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

extern "C" int svt_amd_full_loop_chroma(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdChromaLoopIn *in,
                                        const int16_t *const residual[2], int16_t *const quant[2], int16_t *const recon[2],
                                        uint32_t pitch, SvtAmdChromaLoopOut *out)
{
    return full_loop_chroma_host(ctx, cost, in, residual, quant, recon, pitch, nullptr, out);
}
extern "C" int svt_amd_full_loop_chroma_cabac(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdChromaLoopIn *in,
                                              const int16_t *const residual[2], int16_t *const quant[2], int16_t *const recon[2],
                                              uint32_t pitch, uint32_t *ctx_model, SvtAmdChromaLoopOut *out)
{
    if (!ctx_model)
        return SVT_AMD_ERR_BAD_PARAM;
    return full_loop_chroma_host(ctx, cost, in, residual, quant, recon, pitch, ctx_model, out);
}

/* ------------------------------------------------------------------------------------------------------------------------
 * The encode pass's quantiser at encMode 1..4: UnifiedQuantizeInvQuantize with contextPtr->mdContext->rdoqPmCoreMethod ==
 * EB_PMCORE (Codec/EbTransforms.c:3009-3052) = DecoupledQuantizeInvQuantizeLoops over the whole unit (coefficient shapes,
 * dead-zone override and the clean-ups of the plain path do not apply): regular quantisation, for luma the 4x4-block
 * re-decision above, then every level de-quantised again.  One wave per unit; the unit lives in LDS.
 * ------------------------------------------------------------------------------------------------------------------------ */
struct PmQuantUnit { uint8_t size, qp, bit_depth, slice_type, component, cand_type, pad[2]; uint32_t lambda; }; /* = SvtAmdPmQuantUnit */

__global__ __launch_bounds__(64) void k_pmcore_quant(const SvtAmdCabacCost *__restrict__ cost_p, const PmQuantUnit *__restrict__ units, const int16_t *__restrict__ coeff,
                                                     int16_t *__restrict__ quant, int16_t *__restrict__ recon,
                                                     uint32_t *__restrict__ nzOut)
{
    __shared__ int16_t cf[32 * 32], lev[32 * 32];
    __shared__ int16_t Pq[64][16];
    const SvtAmdCabacCost &c_cost = *cost_p;
    const PmQuantUnit U = units[blockIdx.x];
    const int t = threadIdx.x, N = U.size, lg = 31 - __clz(N);
    const size_t base = (size_t)blockIdx.x * 1024;
    FlUnit Q;
    const int qpRem = U.qp % 6, qpPer = U.qp / 6;
    Q.QF = qpRem == 0 ? 26214u : qpRem == 1 ? 23302u : qpRem == 2 ? 20560u : qpRem == 3 ? 18396u : qpRem == 4 ? 16384u : 14564u;
    const int FFv = qpRem == 0 ? 40 : qpRem == 1 ? 45 : qpRem == 2 ? 51 : qpRem == 3 ? 57 : qpRem == 4 ? 64 : 72;
    const int tshift = 15 - U.bit_depth - lg;
    Q.shiftedQBits = 14 + qpPer + tshift;
    Q.q_offset = ((U.slice_type == 2 || U.slice_type == 3) ? 171u : 85u) << (Q.shiftedQBits - 9);
    Q.shiftedFFunc = qpPer > 8 ? FFv << (qpPer - 2) : FFv << qpPer;
    Q.shiftNum = qpPer > 8 ? 20 - 14 - tshift - 2 : 20 - 14 - tshift;
    Q.iq_offset = 1 << (Q.shiftNum - 1);
    unsigned nz = 0;
    for (int i = t; i < N * N; i += 64) {
        const int v = coeff[base + i], sign = v < 0 ? -1 : 1;
        int tq = abs(v);
        tq = (int)((uint32_t)tq * Q.QF);
        tq = (int)((uint32_t)tq + Q.q_offset);
        tq >>= Q.shiftedQBits;
        const int qv = clip16i(sign * tq);
        cf[i] = (int16_t)v, lev[i] = (int16_t)qv;
        nz += qv != 0;
    }
    for (int o = 32; o > 0; o >>= 1)
        nz += __shfl_xor(nz, o);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (nz != 0 && U.component == 0) {
        pm_core_blocks<64>(c_cost, cf, lev, N, N, lg, t, true, t, (int)U.cand_type, U.lambda, Q, Pq);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    nz = 0;
    for (int i = t; i < N * N; i += 64) {
        const int qv = lev[i];
        quant[base + i] = (int16_t)qv;
        recon[base + i] = (int16_t)clip16i(((qv * Q.shiftedFFunc) + Q.iq_offset) >> Q.shiftNum);
        nz += qv != 0;
    }
    for (int o = 32; o > 0; o >>= 1)
        nz += __shfl_xor(nz, o);
    if (t == 0)
        nzOut[blockIdx.x] = nz;
}

extern "C" int svt_amd_pmcore_quantize_batch(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdPmQuantUnit *d_units,
                                             const int16_t *d_coeff, int16_t *d_quant, int16_t *d_recon, uint32_t *d_nz, uint32_t nunits)
{
    static_assert(sizeof(PmQuantUnit) == sizeof(SvtAmdPmQuantUnit), "unit layout");
    if (!ctx || !cost || !d_units || !d_coeff || !d_quant || !d_recon || !d_nz || !nunits)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rate_upload_tables(ctx, cost);
    if (rc)
        return rc;
    hipLaunchKernelGGL(k_pmcore_quant, dim3(nunits), dim3(64), 0, ctx->stream, (const SvtAmdCabacCost *)ctx->d_cabac_cost, (const PmQuantUnit *)d_units, d_coeff, d_quant, d_recon, d_nz);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

extern "C" int svt_amd_pmcore_quantize(SvtAmdContext *ctx, const SvtAmdCabacCost *cost, const SvtAmdPmQuantUnit *unit, const int16_t *coeff,
                                       uint32_t coeffStride, int16_t *quant, int16_t *recon, uint32_t *nz)
{
    if (!ctx || !cost || !unit || !coeff || !quant || !recon || !nz ||
        !(unit->size == 4 || unit->size == 8 || unit->size == 16 || unit->size == 32) || coeffStride < unit->size ||
        (unit->bit_depth != 8 && unit->bit_depth != 10) || unit->qp > 51)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *d_scratch = nullptr; /* unit | nz | coeff | quant | recon ; callers serialise per context */
    const size_t o_unit = 0, o_nz = 64, o_c = 128, o_q = o_c + 2048, o_r = o_q + 2048, total = o_r + 2048;
    {
        int rc_s = svt_amd_ctx_scratch(ctx, total, &d_scratch);
        if (rc_s)
            return rc_s;
    }
    const int N = unit->size;
    int16_t hc[32 * 32], hq[32 * 32], hr[32 * 32];
    for (int y = 0; y < N; y++)
        ::memcpy(hc + y * N, coeff + (size_t)y * coeffStride, (size_t)N * 2);
    HIP_TRY(hipMemcpyAsync(d_scratch + o_unit, unit, sizeof(*unit), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_scratch + o_c, hc, (size_t)N * N * 2, hipMemcpyHostToDevice, ctx->stream));
    int rc = svt_amd_pmcore_quantize_batch(ctx, cost, (const SvtAmdPmQuantUnit *)(d_scratch + o_unit), (const int16_t *)(d_scratch + o_c),
                                           (int16_t *)(d_scratch + o_q), (int16_t *)(d_scratch + o_r), (uint32_t *)(d_scratch + o_nz), 1);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(nz, d_scratch + o_nz, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hq, d_scratch + o_q, (size_t)N * N * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hr, d_scratch + o_r, (size_t)N * N * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int y = 0; y < N; y++) {
        ::memcpy(quant + (size_t)y * coeffStride, hq + y * N, (size_t)N * 2);
        ::memcpy(recon + (size_t)y * coeffStride, hr + y * N, (size_t)N * 2);
    }
    return SVT_AMD_OK;
}
