/* Device code of the coefficient rate estimator shared by rate_kernels.hip and fullloop_kernels.hip.  The immutable scan /
 * context tables live in per-translation-unit constant memory (uploaded once per device, rate_upload_tables()); the
 * CabacCost_t copy is a device buffer OWNED BY THE CONTEXT (ctx->d_cabac_cost, written stream-ordered on the context's
 * stream) and reaches the device functions as a pointer, so contexts on one device never share it. */
#ifndef SVT_AMD_RATE_DEVICE_H
#define SVT_AMD_RATE_DEVICE_H
#include <mutex>
#include "leaf_util.h"

#define ONE_BIT 32u

struct RateTables {
    uint8_t diag4[16], col4[16], ctx4[3][16], ctx8[2][16];
    uint8_t sb[4][64]; /* sub-block scans by log2(size) - 2; (y << 4) | x */
};
static __constant__ RateTables c_rt;

static void build_tables(RateTables *t)
{
    static const uint8_t ctxIdxMap[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8};
    int n = 0;
    for (int d = 0; d < 7; d++)
        for (int y = d < 4 ? d : 3; y >= 0 && d - y < 4; y--)
            t->diag4[n++] = (uint8_t)(y * 4 + (d - y));
    for (int k = 0; k < 16; k++)
        t->col4[k] = (uint8_t)((k & 3) * 4 + (k >> 2));
    for (int lg = 0; lg < 4; lg++) {
        const int w = lg == 0 ? 2 : 1 << lg;
        n = 0;
        for (int d = 0; d < 2 * w - 1; d++)
            for (int y = d < w ? d : w - 1; y >= 0 && d - y < w; y--)
                t->sb[lg][n++] = (uint8_t)((y << 4) | (d - y));
    }
    for (int k = 0; k < 16; k++) {
        t->ctx4[0][k] = ctxIdxMap[t->diag4[k]];
        t->ctx4[1][k] = ctxIdxMap[k];
        t->ctx4[2][k] = ctxIdxMap[t->col4[k]];
        const int sd = (t->diag4[k] >> 2) + (t->diag4[k] & 3), sr = (k >> 2) + (k & 3);
        t->ctx8[0][k] = sd == 0 ? 2 : sd < 3 ? 1 : 0;
        t->ctx8[1][k] = sr == 0 ? 2 : sr < 3 ? 1 : 0;
    }
}

__device__ __forceinline__ uint32_t golomb_bits0(uint32_t symbol) /* rice parameter 0 */
{
    uint32_t bins = 1;
    if (symbol < 3)
        bins += symbol;
    else
        bins += 2 * (31 - __clz((int)(symbol - 2))) + 3;
    return ONE_BIT * bins;
}
__device__ __forceinline__ uint32_t last_xy_bits(const SvtAmdCabacCost &c_cost, uint32_t x, uint32_t y, uint32_t size, int isChroma)
{
    const int off = (isChroma ? 120 : 0) - 8;
    if (size == 1)
        return c_cost.CabacBitsLast[0] + c_cost.CabacBitsLast[1];
    return c_cost.CabacBitsLast[off + 2 * (int)(x + size)] + c_cost.CabacBitsLast[off + 2 * (int)(y + size) + 1];
}

/* lg = log2(size) in 2..5, S = sub-blocks per TU = lanes per TU */
/* Bits of one TU, computed by the S = (size/4)^2 adjacent lanes that own its sub-blocks (sub = lane & (S-1)); every lane
 * of the wave must call (ballot / shuffles inside).  p: coefficient (0,0) of the TU (global or LDS), row pitch `stride`;
 * live = this lane's TU exists.  The sum lands in all S lanes; the caller shifts it by 10 like the reference. */
__device__ __forceinline__ uint32_t coeff_bits_lanes(const SvtAmdCabacCost &c_cost, const int16_t *p0, uint32_t stride, int lg, const SvtAmdTuInfo ti, bool live,
                                                     int lane, int sub)
{
    const int S = lg == 2 ? 1 : 1 << (2 * (lg - 2));
    const uint32_t size = 1u << lg;
    const int isChroma = ti.component != 0;
    uint32_t scan = 0;
    if (ti.type == 2 && lg <= 3 - isChroma) {
        const uint32_t tc = ti.intra_chroma_mode == 0 ? 0u : ti.intra_chroma_mode == 1 ? 26u : ti.intra_chroma_mode == 2 ? 10u
                            : ti.intra_chroma_mode == 3 ? 1u : 4u;
        const int m = (!isChroma || tc == 4) ? (int)ti.intra_luma_mode : (int)tc;
        if (abs(8 - ((m - 2) & 15)) <= 4)
            scan = (m & 16) ? 1 : 2;
    }
    /* 1. this lane's sub-block in scan order */
    uint32_t lin[16], sig = 0, g1 = 0;
    {
        uint32_t gy = c_rt.sb[lg - 2][sub] >> 4, gx = c_rt.sb[lg - 2][sub] & 15;
        if (scan == 1) { const uint32_t tmp = gx; gx = gy; gy = tmp; }
        const int16_t *p = p0 + 4 * gy * stride + 4 * gx;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t pos = scan ? c_rt.col4[k] : c_rt.diag4[k];
            uint32_t py = pos >> 2, px = pos & 3;
            if (scan == 1) { const uint32_t tmp = px; px = py; py = tmp; }
            const int v = live && ti.num_nonzero ? (int)p[stride * py + px] : 0;
            const uint32_t a = (uint32_t)abs(v);
            lin[k] = a;
            sig |= (uint32_t)(a != 0) << k;
            g1 |= (uint32_t)(a > 1) << k;
        }
    }
    /* 2. last significant sub-block of the TU */
    const unsigned long long seg = (S == 64 ? ~0ull : ((1ull << S) - 1)) << (lane - sub);
    const unsigned long long nzmask = __ballot(sig != 0) & seg;
    const int lastSet = nzmask ? (63 - __clzll((long long)nzmask)) - (lane - sub) : -1;
    uint32_t bits = 0;
    if (lastSet >= 0) {
        const uint32_t first = __shfl(lin[0], lane - sub); /* coeffBufferPtr[0]: position 0 of sub-block 0 in every scan */
        if (ti.num_nonzero == 1 && first != 0) { /* DC-only fast track */
            if (sub == 0) {
                const uint32_t a = first, o1 = isChroma * 16, o2 = isChroma * 4;
                bits = last_xy_bits(c_cost, 0, 0, size, isChroma) + c_cost.CabacBitsG1[2 * (o1 + 1) + (a > 1)];
                if (a > 1) {
                    bits += c_cost.CabacBitsG2[2 * o2 + (a > 2)];
                    if (a > 2)
                        bits += golomb_bits0(a - 3);
                }
                bits += ONE_BIT;
            }
        } else if (sub <= lastSet) {
            const bool isLast = sub == lastSet;
            const int posLast = 31 - __clz((int)(sig | 1));
            if (isLast) { /* position of the last significant coefficient */
                uint32_t ly = 4 * (c_rt.sb[lg - 2][sub] >> 4), lx = 4 * (c_rt.sb[lg - 2][sub] & 15);
                const uint32_t pl = scan ? c_rt.col4[posLast] : c_rt.diag4[posLast];
                ly += pl >> 2, lx += pl & 3;
                if (scan) { const uint32_t tmp = lx; lx = ly; ly = tmp; }
                bits += last_xy_bits(c_cost, lx, ly, size, isChroma);
            }
            bool coded = true;
            if (sub != 0 && !isLast) { /* coded_sub_block_flag */
                bits += c_cost.CabacBitsSigMl[2 * (isChroma * 2) + (sig != 0)];
                coded = sig != 0;
            }
            if (coded) {
                const uint32_t sigOff = isChroma ? 27 : 0;
                uint32_t tOff = 0;
                if (lg != 2) {
                    tOff = lg == 3 ? (scan == 0 ? 9 : 15) : (!isChroma ? 21 : 12);
                    tOff += (!isChroma && sub != 0) ? 3 : 0;
                }
                const uint8_t *bp = c_cost.CabacBitsSig + 2 * sigOff + 2 * tOff;
                const uint32_t cset = (sub != 0 && !isChroma) ? 2 : 0;
                const uint32_t o1 = isChroma * 16 + 4 * cset, o2 = isChroma * 4 + cset;
                /* significance flags are coded for k in [k_low, k_start]; coefficient `inferred` is significant without
                 * a flag (the last position, or the lone DC of a middle sub-block) */
                const bool lone = sig == 1;
                const int inferred = isLast ? posLast : ((lone && sub != 0) ? 0 : -1);
                const int k_start = isLast ? (lone ? -1 : posLast - 1) : 15;
                const int k_low = (sub == 0 || (lone && !isLast)) ? 1 : 0;
                int nnz = 0, phase = 0;
                uint32_t lev = 0;
#pragma unroll
                for (int k = 15; k >= 0; k--) {
                    const uint32_t f = (sig >> k) & 1u;
                    bool take = (k == inferred);
                    if (k <= k_start && k >= k_low) {
                        const uint32_t ci = lg == 2 ? c_rt.ctx4[scan][k] : c_rt.ctx8[scan != 0][k];
                        bits += bp[2 * ci + f];
                        take = take || f;
                    } else if (k == 0 && sub == 0 && !(isLast && lone)) { /* the DC flag has its own context */
                        bits += c_cost.CabacBitsSig[2 * sigOff + f];
                        take = take || f;
                    }
                    if (take) {
                        const uint32_t a = lin[k];
                        if (nnz < 8) {
                            if (phase == 0) {
                                lev += c_cost.CabacBitsG1[2 * (o1 + 1) + (a > 1)];
                                if (a > 1) {
                                    lev += c_cost.CabacBitsG2[2 * o2 + (a > 2)];
                                    if (a > 2)
                                        lev += golomb_bits0(a - 3);
                                    phase = 1;
                                }
                            } else {
                                lev += c_cost.CabacBitsG1[2 * o1 + (a > 1)];
                                if (a > 1)
                                    lev += golomb_bits0(a - 2);
                            }
                        } else {
                            lev += golomb_bits0(a - 1);
                        }
                        nnz++;
                    }
                }
                bits += ONE_BIT * (uint32_t)nnz;
                if (g1 == 0)
                    bits += nnz > 0 ? c_cost.CabacBitsG1x[4 * o1 + nnz - 1] : 0u;
                else
                    bits += lev;
            }
        }
    }
    for (int o = 1; o < S; o <<= 1)
        bits += __shfl_xor(bits, o);
    return bits;
}


/* once per device: the immutable scan / context tables */
static int rate_tables_once(int device)
{
    static std::mutex mu;
    static bool tables_done[64];
    std::lock_guard<std::mutex> g(mu);
    if (device < 0 || device >= 64)
        return SVT_AMD_ERR_BAD_PARAM;
    if (!tables_done[device]) {
        RateTables t;
        build_tables(&t);
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_rt), &t, sizeof(t)));
        tables_done[device] = true;
    }
    return SVT_AMD_OK;
}
/* per call: the caller's CabacCost_t into the context's own buffer, stream-ordered before the kernels that read it */
static int rate_upload_tables(SvtAmdContext *ctx, const SvtAmdCabacCost *cost)
{
    int rc = rate_tables_once(ctx->device);
    if (rc)
        return rc;
    /* pageable source: staged by the runtime before the call returns */
    HIP_TRY(hipMemcpyAsync(ctx->d_cabac_cost, cost, sizeof(*cost), hipMemcpyHostToDevice, ctx->stream));
    return SVT_AMD_OK;
}
#endif
