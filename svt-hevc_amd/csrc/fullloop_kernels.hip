/*
 * Luma and chroma full loops of mode-decision candidates, fused (EncDec row "PerformFullLoop, ProductFullLoop ...").
 *
 * Replaces ProductFullLoop (Codec/EbFullLoop.c:185-446) and the pair FullLoop_R + CuFullDistortionFastTuMode_R (:579-1066)
 * for the presets' common configuration (no RDOQ / PM-core, coefficient-domain distortion, no CABAC-context update).
 * A workgroup owns GB transform units at a time: one 32x32 or 16x16 (four waves), four 8x8 (one wave: four luma
 * candidates, or Cb + Cr of two), sixteen 4x4 (one wave: Cb + Cr of eight candidates); a 64x64 CU walks its four units
 * one after the other.  Everything between the residual and the cost decision
 * stays on chip:
 *   residual -> LDS -> forward "Estimate" DCT (two passes in LDS, txfm_device.h)
 *            -> quantise / inverse-quantise the (T >> pf) area, count non-zeros, coefficient-domain distortions
 *            -> coefficient bits (one lane per 4x4 sub-block, rate_device.h) -> cbf / cost decision (one thread).
 * Only the quantised and reconstructed coefficients and a 64-byte result record leave the workgroup.
 * HBM traffic per TU: 2 B/sample in, 4 B/sample out (area only) - the five-kernel composition moves 14 B/sample.
 */
#include "txfm_device.h"
#include "rate_device.h"
#include <cstring>

/* one transform unit in flight */
struct FlSlot {
    int active;            /* candidate exists and has this launch's transform size */
    uint32_t cand, plane;  /* candidate index; 0 luma / Cb, 1 Cr */
    uint32_t base, pitch;  /* sample offset of the unit's (0,0) in the residual / quant / recon arrays, row pitch */
    int area, lg;          /* quantised area (T >> pf) and its log2 */
    uint32_t QF, q_offset;
    int shiftedQBits, shiftedFFunc, iq_offset, shiftNum;
    int cand_type, intra_luma_mode;
};

template <int N, int G>
struct FlShared {
    static constexpr int GB = G;
    int16_t q[GB][N * N];          /* quantised coefficients of the units in flight, row pitch N */
    unsigned nz[GB], res[GB], pred[GB], bits[GB];
    FlSlot slot[GB];
    /* running results of the slot's candidate (plane), kept here rather than in registers of every thread */
    struct Acc {
        unsigned long long bits, d0, d1;
        uint32_t cbf, nzs[5];
        int16_t ydc[4];
        uint16_t cand_nz[4];
    } acc[GB];
};

__device__ __forceinline__ void fl_quant_params(FlSlot &S, uint32_t qp, uint32_t slice_type, int LG)
{
    /* ProductUnifiedQuantizeInvQuantizeMd (EbFullLoop.c:98-113) = UnifiedQuantizeInvQuantize_R (:483-497) at bit depth 8 */
    const int qpRem = (int)(qp % 6), qpPer = (int)(qp / 6);
    S.QF = qpRem == 0 ? 26214u : qpRem == 1 ? 23302u : qpRem == 2 ? 20560u : qpRem == 3 ? 18396u : qpRem == 4 ? 16384u : 14564u;
    const int FFv = qpRem == 0 ? 40 : qpRem == 1 ? 45 : qpRem == 2 ? 51 : qpRem == 3 ? 57 : qpRem == 4 ? 64 : 72;
    const int tshift = 7 - LG;
    S.shiftedQBits = 14 + qpPer + tshift;
    S.q_offset = ((slice_type == 2 || slice_type == 3) ? 171u : 85u) << (S.shiftedQBits - 9);
    S.shiftedFFunc = qpPer > 8 ? FFv << (qpPer - 2) : FFv << qpPer;
    S.shiftNum = qpPer > 8 ? 20 - 14 - tshift - 2 : 20 - 14 - tshift;
    S.iq_offset = 1 << (S.shiftNum - 1);
}

/* N: transform size of this launch.  CHROMA 0: SvtAmdFullLoopIn/Out, residual slab 4096 samples per candidate;
 * CHROMA 1: SvtAmdChromaLoopIn/Out, slab 2048 (Cb then Cr).  Workgroup b serves the candidates
 * [b*CPW, (b+1)*CPW) (CPW = candidates per workgroup); for the one-unit-at-a-time sizes of the chroma loop
 * blockIdx.y is the plane. */
template <int N, bool CHROMA, int NT, int GB>
__global__ __launch_bounds__(NT) void k_full_loop(const void *__restrict__ in_all, const int16_t *__restrict__ residual,
                                                  int16_t *__restrict__ quant, int16_t *__restrict__ recon,
                                                  void *__restrict__ out_all, uint32_t ncand, int shift1, int shift2,
                                                  int wrap_levels)
{
    constexpr int LG = N == 32 ? 5 : N == 16 ? 4 : N == 8 ? 3 : 2;
    constexpr int PLANES = CHROMA ? 2 : 1;
    constexpr int CPW = GB >= PLANES ? GB / PLANES : 1; /* candidates per workgroup */
    __shared__ TxShared<N, GB> X;
    __shared__ FlShared<N, GB> F;
    const int t = threadIdx.x;
    /* lane s owns slot s; with one unit in flight every thread mirrors slot 0 */
    int my_size = 0, my_active = 0;
    uint32_t my_cand = 0, my_plane = 0;
    if (CPW == 1 || t < GB) { /* one candidate per workgroup: every thread mirrors it (slot t % GB) */
        const int s = t % GB;
        my_cand = blockIdx.x * CPW + (uint32_t)s / PLANES;
        my_plane = GB >= PLANES ? (uint32_t)s % PLANES : blockIdx.y;
        if (my_cand < ncand) {
            if (CHROMA) {
                const SvtAmdChromaLoopIn *in = (const SvtAmdChromaLoopIn *)in_all + my_cand;
                my_size = (int)in->size;
                my_active = (my_size == 64 ? 16 : my_size >> 1) == N;
            } else {
                const SvtAmdFullLoopIn *in = (const SvtAmdFullLoopIn *)in_all + my_cand;
                my_size = (int)in->size;
                my_active = (my_size == 64 ? 32 : my_size) == N;
            }
        }
    }
    /* this launch serves one transform size: leave at once when none of the workgroup's candidates has it
     * (uniform without a barrier: one unit -> every thread looked at the same candidate; several -> a single wave) */
    if (CPW == 1) {
        if (!my_active)
            return;
    } else if (NT == 64) {
        if (!__ballot(my_active))
            return;
    } else { /* several units, several waves: agree through LDS */
        __shared__ int s_any;
        if (t == 0)
            s_any = 0;
        __syncthreads();
        if (my_active)
            s_any = 1;
        __syncthreads();
        if (!s_any)
            return;
    }
    /* a 64x64 CU has four units; only the one-unit-at-a-time sizes can meet one */
    const int ntu = (CPW == 1 && my_size == 64) ? 4 : 1;
    for (int i = t; i < 32 * 32; i += NT)
        (&X.T[0][0])[i] = (&c_T32[0][0])[i];
    if (t < GB) {
        typename FlShared<N, GB>::Acc a;
        a.bits = 0, a.d0 = 0, a.d1 = 0, a.cbf = 0;
        for (int k = 0; k < 5; k++)
            a.nzs[k] = 0;
        for (int k = 0; k < 4; k++)
            a.ydc[k] = 0, a.cand_nz[k] = 0;
        if (!CHROMA && my_active) {
            const SvtAmdFullLoopIn *in = (const SvtAmdFullLoopIn *)in_all + my_cand;
            a.cbf = in->ycbf, a.bits = in->coeff_bits;
            a.d0 = my_size == 64 ? in->dist[0] : 0, a.d1 = my_size == 64 ? in->dist[1] : 0;
        }
        F.acc[t] = a;
    }

    for (int tu = 0; tu < ntu; tu++) {
        if (t < GB) {
            FlSlot S;
            S.active = my_active, S.cand = my_cand, S.plane = my_plane;
            S.area = N, S.lg = LG, S.base = 0, S.pitch = N, S.cand_type = 0, S.intra_luma_mode = 0;
            S.QF = 0, S.q_offset = 0, S.shiftedQBits = 0, S.shiftedFFunc = 0, S.iq_offset = 1, S.shiftNum = 1;
            if (my_active) {
                uint32_t pf, qp, slice;
                if (CHROMA) {
                    const SvtAmdChromaLoopIn *in = (const SvtAmdChromaLoopIn *)in_all + my_cand;
                    /* correctedPFMode (EbFullLoop.c:647-652): 4x4 never, 8x8 at most N2 */
                    pf = N == 4 ? 0u : (N == 8 && in->pf_mode == 2 ? 1u : in->pf_mode);
                    qp = my_plane ? in->cr_qp : in->cb_qp, slice = in->slice_type;
                    S.cand_type = (int)in->cand_type, S.intra_luma_mode = (int)in->intra_luma_mode;
                    S.pitch = (uint32_t)my_size >> 1;
                    S.base = my_cand * 2048u + my_plane * 1024u + (my_size == 64 ? ((tu & 1) << 4) + (tu > 1 ? 16 * 32 : 0) : 0);
                } else {
                    const SvtAmdFullLoopIn *in = (const SvtAmdFullLoopIn *)in_all + my_cand;
                    pf = in->pf_mode, qp = in->qp, slice = in->slice_type;
                    S.cand_type = (int)in->cand_type, S.intra_luma_mode = (int)in->intra_luma_mode;
                    S.pitch = (uint32_t)my_size;
                    S.base = my_cand * 4096u + (my_size == 64 ? ((tu & 1) << 5) + (tu > 1 ? 32 * 64 : 0) : 0);
                }
                S.area = N >> pf, S.lg = LG - (int)pf;
                fl_quant_params(S, qp, slice, LG);
            }
            F.slot[t] = S;
            F.nz[t] = 0, F.res[t] = 0, F.pred[t] = 0, F.bits[t] = 0;
        }
        __syncthreads();
        for (int i = t; i < GB * N * N; i += NT) {
            const int s = i / (N * N), e = i - s * (N * N);
            const FlSlot &S = F.slot[s];
            (&X.io[0][0])[i] = S.active ? residual[S.base + (e / N) * S.pitch + (e % N)] : (int16_t)0;
        }
        __syncthreads();
        fwd_pass<N, false, NT, GB>(X, shift1, wrap_levels, nullptr, GB, t);
        fwd_pass<N, false, NT, GB>(X, shift2, wrap_levels, nullptr, GB, t);
        /* QuantizeInvQuantize (C_DEFAULT/EbTransforms_C.c:89) over the area + the two coefficient-domain sums */
        unsigned nz1 = 0, res1 = 0, pred1 = 0;
        for (int i = t; i < GB * N * N; i += NT) {
            const int s = i / (N * N), e = i - s * (N * N), r = e / N, c = e - r * N;
            const FlSlot &S = F.slot[s];
            if (!S.active || r >= S.area || c >= S.area)
                continue;
            const int v = X.io[s][e], sign = v < 0 ? -1 : 1;
            int tq = abs(v);
            tq = (int)((uint32_t)tq * S.QF);
            tq = (int)((uint32_t)tq + S.q_offset);
            tq >>= S.shiftedQBits;
            const int qv = clip16i(sign * tq);
            const int rv = clip16i(((qv * S.shiftedFFunc) + S.iq_offset) >> S.shiftNum);
            F.q[s][e] = (int16_t)qv;
            quant[S.base + r * S.pitch + c] = (int16_t)qv;
            recon[S.base + r * S.pitch + c] = (int16_t)rv;
            const int16_t d = (int16_t)(v - rv);
            if (GB == 1) {
                nz1 += qv != 0, res1 += (unsigned)(d * d), pred1 += (unsigned)(v * v);
            } else {
                if (qv)
                    atomicAdd(&F.nz[s], 1u);
                atomicAdd(&F.res[s], (unsigned)(d * d));
                atomicAdd(&F.pred[s], (unsigned)(v * v));
            }
        }
        if (GB == 1) { /* one unit: reduce inside the wave first */
            for (int o = 32; o > 0; o >>= 1)
                nz1 += __shfl_xor(nz1, o), res1 += __shfl_xor(res1, o), pred1 += __shfl_xor(pred1, o);
            if ((t & 63) == 0) {
                atomicAdd(&F.nz[0], nz1);
                atomicAdd(&F.res[0], res1);
                atomicAdd(&F.pred[0], pred1);
            }
        }
        __syncthreads();
        /* TuEstimateCoeffBitsLuma / TuEstimateCoeffBits_R: one lane per 4x4 sub-block; units of equal area together */
#pragma unroll 1
        for (int pfv = 0; pfv < ((N == 4 || GB == 1) ? 1 : 3) && t < 64; pfv++) { /* wave 0 */
            const int lg = GB == 1 ? __builtin_amdgcn_readfirstlane(F.slot[0].lg) : LG - pfv;
            if (lg < 2)
                break;
            const int S4 = lg == 2 ? 1 : 1 << (2 * (lg - 2));
            const int s = t / S4, sub = t - s * S4;
            bool live = s < GB;
            SvtAmdTuInfo ti = {0, 0, 0, 4 /* EB_INTRA_CHROMA_DM */, 0};
            const int16_t *p0 = &F.q[0][0];
            if (GB == 1) {
                /* one unit: everything about it is wave-uniform - keep it in scalar registers */
                const FlSlot &S = F.slot[0];
                const int act = __builtin_amdgcn_readfirstlane(S.active && S.lg == lg && F.nz[0] != 0);
                live = live && act;
                ti.num_nonzero = (uint32_t)__builtin_amdgcn_readfirstlane((int)F.nz[0]);
                ti.type = (uint8_t)__builtin_amdgcn_readfirstlane(S.cand_type);
                ti.intra_luma_mode = (uint8_t)__builtin_amdgcn_readfirstlane(S.intra_luma_mode);
                ti.component = CHROMA ? (uint8_t)(__builtin_amdgcn_readfirstlane((int)S.plane) + 1) : (uint8_t)0;
            } else if (live) {
                const FlSlot &S = F.slot[s];
                live = S.active && S.lg == lg && F.nz[s] != 0;
                if (live) {
                    ti.num_nonzero = F.nz[s], ti.type = (uint8_t)S.cand_type, ti.intra_luma_mode = (uint8_t)S.intra_luma_mode;
                    ti.component = CHROMA ? (uint8_t)(S.plane + 1) : (uint8_t)0;
                    p0 = &F.q[s][0];
                }
            }
            if (!__ballot(live))
                continue;
            const uint32_t b = coeff_bits_lanes(p0, N, lg, ti, live, t, sub);
            if (live && sub == 0)
                F.bits[s] = b;
        }
        __syncthreads();
        if (t < GB && my_active) {
            const FlSlot &S = F.slot[t];
            const unsigned tnz = F.nz[t];
            /* PictureFullDistortionLuma / _R table [nz != 0][intra] + the scaling of the caller */
            const int mode = tnz == 0 ? 1 : (S.cand_type == 2 ? 2 : 0);
            unsigned long long d0 = mode == 1 ? F.pred[t] : F.res[t], d1 = mode == 2 ? F.res[t] : F.pred[t];
            const int dshift = (!CHROMA && my_size == 64) ? 4 : 2 * (7 - LG);
            d0 = (d0 + (1ull << (dshift - 1))) >> dshift;
            d1 = (d1 + (1ull << (dshift - 1))) >> dshift;
            const unsigned long long tuBits = ((unsigned long long)F.bits[t] << 10) >> 15;
            const int tuIndex = my_size == 64 ? tu + 1 : 0;
            typename FlShared<N, GB>::Acc &A = F.acc[t];
            A.nzs[tuIndex] = tnz;
            if (CHROMA) {
                /* TuCalcCost, chroma branches (EbRateDistortionCost.c:273-279) */
                A.cbf |= (uint32_t)(tnz != 0) << tuIndex;
                A.bits += tuBits, A.d0 += d0, A.d1 += d1;
            } else {
                /* TuCalcCostLuma (EbRateDistortionCost.c:289) */
                const SvtAmdFullLoopIn *in = (const SvtAmdFullLoopIn *)in_all + my_cand;
                const int ctx = my_size == N;
                const unsigned long long nzRate = (tuBits << 15) + in->cbf_bits[2 + ctx], zRate = in->cbf_bits[ctx];
                const unsigned long long lam = in->full_lambda;
                const unsigned long long zCost = S.cand_type == 2 ? ~0ull : (d1 << 8) + (((lam * zRate) + (1u << 22)) >> 23);
                const unsigned long long nzCost = (d0 << 8) + (((lam * nzRate) + (1u << 22)) >> 23);
                const bool keep = nzCost < zCost;
                A.cbf |= (uint32_t)((tnz != 0) && keep) << tuIndex;
                A.bits += keep ? tuBits : 0;
                A.d0 += keep ? d0 : d1;
                A.d1 += d1;
                A.ydc[my_size == 64 ? tu : 0] = (int16_t)abs((int)F.q[t][0]);
                A.cand_nz[my_size == 64 ? tu : 0] = (uint16_t)tnz;
            }
        }
        __syncthreads();
    }
    if (t < GB && my_active) {
        const typename FlShared<N, GB>::Acc &A = F.acc[t];
        if (CHROMA) {
            SvtAmdChromaLoopOut *o = (SvtAmdChromaLoopOut *)out_all + my_cand;
            for (int k = 0; k < 5; k++)
                o->nz[my_plane][k] = A.nzs[k];
            o->cbf[my_plane] = A.cbf, o->coeff_bits[my_plane] = A.bits, o->dist[my_plane][0] = A.d0, o->dist[my_plane][1] = A.d1;
        } else {
            SvtAmdFullLoopOut o;
            for (int k = 0; k < 5; k++)
                o.nz[k] = A.nzs[k];
            for (int k = 0; k < 4; k++)
                o.ydc[k] = A.ydc[k], o.cand_nz[k] = A.cand_nz[k];
            o.ycbf = A.cbf, o.coeff_bits = A.bits, o.dist[0] = A.d0, o.dist[1] = A.d1;
            ((SvtAmdFullLoopOut *)out_all)[my_cand] = o;
        }
    }
}

template <int N, bool CHROMA>
static void launch_full_loop(hipStream_t st, const void *d_in, const int16_t *d_res, int16_t *d_q, int16_t *d_r, void *d_out,
                             uint32_t ncand, int s1, int s2, int wrap)
{
    /* units in flight per workgroup and its width: four waves share one 32x32 unit, the Cb + Cr 16x16 units of one
     * candidate (a 64x64 CU has four such pairs in a row) or four luma 16x16 units; the smaller sizes run one wave over
     * 4 / 16 units */
    constexpr int GB = N == 32 ? 1 : N == 16 ? (CHROMA ? 2 : 4) : N == 8 ? 4 : 16;
    constexpr int NT = N >= 16 ? 256 : 64;
    constexpr int PLANES = CHROMA ? 2 : 1, CPW = GB >= PLANES ? GB / PLANES : 1;
    const dim3 grid((ncand + CPW - 1) / CPW, GB >= PLANES ? 1 : PLANES);
    hipLaunchKernelGGL((k_full_loop<N, CHROMA, NT, GB>), grid, dim3(NT), 0, st, d_in, d_res, d_q, d_r, d_out, ncand, s1, s2, wrap);
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
    /* one launch per transform size; a workgroup none of whose candidates has that size returns at once (sorting the
     * batch by CU size keeps the 8x8 workgroups full).
     * EstimateTransform shifts: Transform32x32Estimate 6/9 wrap 2, Transform16x16Estimate 4/9 wrap 1, Transform8x8 2/9 */
    launch_full_loop<32, false>(ctx->stream, d_in, d_residual, d_quant, d_recon, d_out, ncand, 6, 9, 2);
    launch_full_loop<16, false>(ctx->stream, d_in, d_residual, d_quant, d_recon, d_out, ncand, 4, 9, 1);
    launch_full_loop<8, false>(ctx->stream, d_in, d_residual, d_quant, d_recon, d_out, ncand, 2, 9, 0);
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
    int rc = rate_upload_tables(cost, ctx->stream);
    if (rc)
        return rc;
    /* EstimateTransform shifts: Transform16x16Estimate 4/9 wrap 1, Transform8x8 2/9, Transform4x4 1/8 */
    launch_full_loop<16, true>(ctx->stream, d_in, d_residual, d_quant, d_recon, d_out, ncand, 4, 9, 1);
    launch_full_loop<8, true>(ctx->stream, d_in, d_residual, d_quant, d_recon, d_out, ncand, 2, 9, 0);
    launch_full_loop<4, true>(ctx->stream, d_in, d_residual, d_quant, d_recon, d_out, ncand, 1, 8, 0);
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
