/* Quantiser parameters of a transform unit and the 4x4-block re-decision of the PM-core quantiser (encMode 1..4), shared by the mode
 * decision's fused full loops (fullloop_kernels.hip) and the device-resident encode pass (encdec_kernels.hip). */
#ifndef SVT_AMD_PMCORE_DEVICE_H
#define SVT_AMD_PMCORE_DEVICE_H
#include "txfm_device.h"
#include "rate_device.h"

/* Per-unit parameters, held by every lane of the unit (lane r of the unit = row r of the residual in the first transform
 * pass, column r of the coefficient block afterwards). */
struct FlUnit {
    int active;            /* candidate exists, has this launch's transform size and still has units left */
    uint32_t base, pitch;  /* sample offset of the unit's (0,0) in the residual / quant / recon arrays, row pitch */
    int area, lg;          /* quantised area (T >> pf) and its log2 */
    uint32_t QF, q_offset;
    int shiftedQBits, shiftedFFunc, iq_offset, shiftNum;
};

__device__ __forceinline__ void fl_quant_params(FlUnit &S, uint32_t qp, uint32_t slice_type, int LG)
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

/* The 4x4-block re-decision of DecoupledQuantizeInvQuantizeLoops' EB_PMCORE branch (Codec/EbTransforms.c:2807-2950) for one
 * unit whose coefficients (cf_lds) and regular levels (lev_lds) lie in LDS with row pitch `pitch`: lane r of the unit's
 * STRIDE lanes takes blocks r, r + STRIDE, ...; every block that holds a level is re-quantised from its coefficients scaled by
 * 100 / 70 / 50 % (MatMultOut :39-71; the DC of block 0 passes unscaled when its regular level exceeds PM_DC_TRSHLD1) and the
 * cheapest of the three in coefficient-domain SSE + lambda * (4x4 rate estimate) replaces it in lev_lds.  All lanes of the
 * wave call together (pmu = this lane's unit takes part); Pq = 16 samples of LDS scratch per lane of the wave. */
template <int STRIDE>
__device__ __forceinline__ void pm_core_blocks(const SvtAmdCabacCost &c_cost, const int16_t *cf_lds, int16_t *lev_lds, int pitch, int area, int lg_area, int r, bool pmu,
                                               int t, int cand_type, uint32_t full_lambda, const FlUnit &Q, int16_t (*Pq)[16])
{
    const int nb = area >> 2, nblk = pmu ? nb * nb : 0;
    int nblk_max = nblk;
    for (int o = 32; o > 0; o >>= 1)
        nblk_max = max(nblk_max, __shfl_xor(nblk_max, o));
    const int sse_shift = 2 * (7 - lg_area);
#pragma unroll 1
    for (int b0 = 0; b0 < nblk_max; b0 += STRIDE) {
        const int b = b0 + r;
        const bool liveb = b < nblk;
        const int by = liveb ? b / nb : 0, bx = liveb ? b - by * nb : 0;
        const int off = by * 4 * pitch + bx * 4;
        int cf[16];
        bool any = false;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            cf[k] = liveb ? (int)cf_lds[off + (k >> 2) * pitch + (k & 3)] : 0;
            any = any || (liveb && lev_lds[off + (k >> 2) * pitch + (k & 3)] != 0);
        }
        const bool dc_pass = any && b == 0 && abs((int)lev_lds[0]) > 10;
        unsigned long long best = 0xFFFFFFFFFFFFFFull; /* MAX_CU_COST */
        uint32_t bq[8]; /* the best candidate's 16 levels, packed in pairs */
#pragma unroll
        for (int k = 0; k < 8; k++)
            bq[k] = 0;
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            const int m = c == 0 ? 256 : c == 1 ? 179 : 128;
            unsigned nzc = 0, sres = 0, spred = 0;
            uint32_t pk[8];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int v = cf[k];
                int tr = (abs(v) * m + 128) >> 8;
                tr = clip16i(v < 0 ? -tr : tr);
                if (k == 0 && dc_pass)
                    tr = v;
                int tq = (int)((uint32_t)abs(tr) * Q.QF);
                tq = (int)((uint32_t)tq + Q.q_offset);
                tq >>= Q.shiftedQBits;
                const int qv = clip16i(tr < 0 ? -tq : tq);
                const int rv = clip16i(((qv * Q.shiftedFFunc) + Q.iq_offset) >> Q.shiftNum);
                const int16_t d = (int16_t)(v - rv);
                nzc += qv != 0, sres += (unsigned)(d * d), spred += (unsigned)(v * v);
                Pq[t][k] = (int16_t)qv;
                if (k & 1)
                    pk[k >> 1] |= (uint32_t)(uint16_t)qv << 16;
                else
                    pk[k >> 1] = (uint32_t)(uint16_t)qv;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const bool price = any && nzc != 0;
            SvtAmdTuInfo ti = {price ? nzc : 0u, (uint8_t)cand_type, 0, 0, 0};
            const uint32_t b32 = coeff_bits_lanes(c_cost, &Pq[t][0], 4, 2, ti, price, t, 0);
            unsigned long long sse = nzc ? sres : spred;
            sse = (sse + (1ull << (sse_shift - 1))) >> sse_shift;
            const unsigned long long bits = price ? (unsigned long long)b32 << 10 : 0ull;
            const unsigned long long cst = (sse << 8) + (((unsigned long long)full_lambda * bits + (1u << 22)) >> 23);
            if (cst < best) {
                best = cst;
#pragma unroll
                for (int k = 0; k < 8; k++)
                    bq[k] = pk[k];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (any) {
#pragma unroll
            for (int k = 0; k < 16; k++)
                lev_lds[off + (k >> 2) * pitch + (k & 3)] = (int16_t)((k & 1) ? bq[k >> 1] >> 16 : bq[k >> 1] & 0xffffu);
        }
    }
}

#endif
