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
    uint8_t ctx8p[2][4][16]; /* sigCtx by (scan != diagonal, prevCsbf pattern, scan position): H.265 9.3.4.2.5 (context-updating estimator) */
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
    for (int m = 0; m < 2; m++)
        for (int pat = 0; pat < 4; pat++)
            for (int k = 0; k < 16; k++) {
                const int pos = m ? t->col4[k] : t->diag4[k], yP = pos >> 2, xP = pos & 3;
                t->ctx8p[m][pat][k] = (uint8_t)(pat == 0 ? (xP + yP == 0 ? 2 : xP + yP < 3 ? 1 : 0)
                                                : pat == 1 ? (yP == 0 ? 2 : yP == 1 ? 1 : 0)
                                                : pat == 2 ? (xP == 0 ? 2 : xP == 1 ? 1 : 0) : 2);
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
                                                     int lane, int sub, const RateTables &rt = c_rt /* the scan / context tables: the constant-memory copy, or the caller's in LDS */)
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
        uint32_t gy = rt.sb[lg - 2][sub] >> 4, gx = rt.sb[lg - 2][sub] & 15;
        if (scan == 1) { const uint32_t tmp = gx; gx = gy; gy = tmp; }
        const int16_t *p = p0 + 4 * gy * stride + 4 * gx;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t pos = scan ? rt.col4[k] : rt.diag4[k];
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
                uint32_t ly = 4 * (rt.sb[lg - 2][sub] >> 4), lx = 4 * (rt.sb[lg - 2][sub] & 15);
                const uint32_t pl = scan ? rt.col4[posLast] : rt.diag4[posLast];
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
                /* the highest position any lane of the wave has something to price at (its last significant coefficient in the TU's last sub-block, 15 in a coded
                 * sub-block before it), wave-uniform: the positions above are skipped by a scalar branch.  A TU whose levels sit in the first few positions of the DC
                 * sub-block - the common case of the mode decision's full loops - walks those few, not sixteen. */
                int kmax;
                {
                    const int my_top = isLast ? posLast : 15;
                    kmax = __ballot(my_top >= 8) != 0 ? 8 : 0;
                    kmax += __ballot(my_top >= kmax + 4) != 0 ? 4 : 0;
                    kmax += __ballot(my_top >= kmax + 2) != 0 ? 2 : 0;
                    kmax += __ballot(my_top >= kmax + 1) != 0 ? 1 : 0;
                }
#pragma unroll
                for (int k = 15; k >= 0; k--) {
                    if (k > kmax)
                        continue;
                    const uint32_t f = (sig >> k) & 1u;
                    bool take = (k == inferred);
                    if (k <= k_start && k >= k_low) {
                        const uint32_t ci = lg == 2 ? rt.ctx4[scan][k] : rt.ctx8[scan != 0][k];
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



/* =====================================================================================================================
 * The CABAC-context-UPDATING estimator (EstimateQuantizedCoefficients_generic_Update, Codec/EbEntropyCoding.c:2986-3480, with
 * EstimateLastSignificantXY_UPDATE :2374 and EstimateRemainingCoeffExponentialGolombCode, EbEntropyCodingUtil.c:326): every
 * context-coded bin costs c_ebits[bin ^ state] and moves the state of its context model (UPDATE_CONTEXT_MODEL,
 * EbMdRateEstimation.h:118).  The chain of states is sequential by definition (each bin's price depends on every earlier bin of
 * the same context, and the mode decision threads the model from unit to unit), so ONE lane walks a unit; its neighbours first
 * build the per-sub-block significance maps in parallel (rate_update_sigmaps).  The model lives in LDS as 136 bytes in
 * CoeffCtxtMdl_t's order (EbCabacContextModel.h:204-214; 7-bit states).
 * c_next is H.265 Table 9-41 in the reference's packing ((pStateIdx << 1) | valMps; [0..127] bin == MPS, [128..255] LPS);
 * c_ebits are HM's ContextModel::m_entropyBits as the reference carries them (Codec/EbHmCode.c:216-246): data.
 * ===================================================================================================================== */
#define RATE_CTX_WORDS 136
#define UP_ONE_BIT 32768u
enum { CX_LASTX = 0, CX_LASTY = 30, CX_SIG = 60, CX_CG = 102, CX_G1 = 106, CX_G2 = 130 };
static __constant__ uint32_t c_ebits[128] = {
    0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
    0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
    0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
    0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
    0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
    0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
    0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
    0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb};
static __constant__ uint8_t c_next[256] = {
    2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
    34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65,
    66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95, 96, 97,
    98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124, 125, 124, 125, 126, 127,
    1, 0, 0, 1, 2, 3, 4, 5, 4, 5, 8, 9, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 18, 19, 22, 23, 22, 23, 24, 25,
    26, 27, 26, 27, 30, 31, 30, 31, 32, 33, 32, 33, 36, 37, 36, 37, 38, 39, 38, 39, 42, 43, 42, 43, 44, 45, 44, 45, 46, 47, 48, 49,
    48, 49, 50, 51, 52, 53, 52, 53, 54, 55, 54, 55, 56, 57, 58, 59, 58, 59, 60, 61, 60, 61, 60, 61, 62, 63, 64, 65, 64, 65, 66, 67,
    66, 67, 66, 67, 68, 69, 68, 69, 70, 71, 70, 71, 70, 71, 72, 73, 72, 73, 72, 73, 74, 75, 74, 75, 74, 75, 76, 77, 76, 77, 126, 127,
};

__device__ __forceinline__ uint32_t up_bin(uint8_t *m, uint32_t bin)
{
    const uint32_t st = *m, c = c_ebits[bin ^ st];
    *m = c_next[((bin ^ (st & 1u)) << 7) | st];
    return c;
}
__device__ __forceinline__ uint32_t up_golomb(uint32_t symbol, uint32_t param)
{
    int cw = (int)(symbol >> param);
    uint32_t bins = param + 1;
    if (cw < 3)
        bins += (uint32_t)cw;
    else
        bins += 2 * (31 - __clz(cw - 2)) + 3;
    return UP_ONE_BIT * bins;
}
/* mode-dependent scan of an intra unit (0 diagonal, 1 horizontal, 2 vertical) */
__device__ __forceinline__ uint32_t rate_scan_of(int lg, const SvtAmdTuInfo ti)
{
    const int isChroma = ti.component != 0;
    uint32_t scan = 0;
    if (ti.type == 2 && lg <= 3 - isChroma) {
        const uint32_t tc = ti.intra_chroma_mode == 0 ? 0u : ti.intra_chroma_mode == 1 ? 26u : ti.intra_chroma_mode == 2 ? 10u
                            : ti.intra_chroma_mode == 3 ? 1u : 4u;
        const int m = (!isChroma || tc == 4) ? (int)ti.intra_luma_mode : (int)tc;
        if (abs(8 - ((m - 2) & 15)) <= 4)
            scan = (m & 16) ? 1 : 2;
    }
    return scan;
}
/* |coefficient| at scan position k of sub-block `sub` */
__device__ __forceinline__ uint32_t up_abs_at(const int16_t *p0, uint32_t stride, int lg, uint32_t scan, int sub, int k)
{
    uint32_t gy = c_rt.sb[lg - 2][sub] >> 4, gx = c_rt.sb[lg - 2][sub] & 15;
    if (scan == 1) { const uint32_t tmp = gx; gx = gy; gy = tmp; }
    const uint32_t pos = scan ? c_rt.col4[k] : c_rt.diag4[k];
    uint32_t py = pos >> 2, px = pos & 3;
    if (scan == 1) { const uint32_t tmp = px; px = py; py = tmp; }
    const int v = p0[(4 * gy + py) * stride + 4 * gx + px];
    return (uint32_t)(v < 0 ? -v : v);
}
/* significance map of sub-block `sub` in scan order (bit k = position k) */
__device__ __forceinline__ uint32_t up_sigmap(const int16_t *p0, uint32_t stride, int lg, uint32_t scan, int sub)
{
    uint32_t sig = 0;
#pragma unroll 4
    for (int k = 0; k < 16; k++)
        sig |= (uint32_t)(up_abs_at(p0, stride, lg, scan, sub, k) != 0) << k;
    return sig;
}
/* The walk: one lane.  M: the model (136 bytes, LDS), sigm: the unit's significance maps by sub-block (LDS), absC: 16 halfwords
 * of LDS scratch.  p0: coefficient (0,0) of the (size x size) area, row pitch `stride`.  nnz must be > 0.
 * Returns what the reference adds to *coeffBitsLong (15 fractional bits). */
__device__ uint32_t coeff_bits_update_walk(uint8_t *M, const uint16_t *sigm, uint16_t *absC, const int16_t *p0, uint32_t stride, int lg,
                                           const SvtAmdTuInfo ti)
{
    const int isChroma = ti.component != 0;
    const uint32_t size = 1u << lg;
    uint32_t bits = 0;
    if (ti.num_nonzero == 1 && p0[0] != 0) { /* DC-only fast track :3064 */
        const int off = isChroma ? 15 : (lg - 2) * 3 + ((lg - 1) >> 2);
        const int a = abs((int)p0[0]);
        bits += up_bin(&M[CX_LASTX + off], 0);
        bits += up_bin(&M[CX_LASTY + off], 0);
        bits += up_bin(&M[CX_G1 + isChroma * 16 + 1], a > 1);
        if (a > 1) {
            bits += up_bin(&M[CX_G2 + isChroma * 4], a > 2);
            if (a > 2)
                bits += up_golomb((uint32_t)a - 3, 0);
        }
        return bits + UP_ONE_BIT;
    }
    const uint32_t scan = rate_scan_of(lg, ti);
    const int nsub = lg == 2 ? 1 : 1 << (2 * (lg - 2));
    int lastSet = 0;
    for (int sb = nsub - 1; sb >= 0; sb--)
        if (sigm[sb]) {
            lastSet = sb;
            break;
        }
    const uint32_t posLast = 31 - __clz((int)sigm[lastSet]);
    {   /* EstimateLastSignificantXY_UPDATE :2374 */
        uint32_t ly = 4 * (c_rt.sb[lg - 2][lastSet] >> 4), lx = 4 * (c_rt.sb[lg - 2][lastSet] & 15);
        const uint32_t pl = scan ? c_rt.col4[posLast] : c_rt.diag4[posLast];
        ly += pl >> 2, lx += pl & 3;
        if (scan) { const uint32_t tmp = lx; lx = ly; ly = tmp; }
        const int off = isChroma ? 15 : (lg - 2) * 3 + ((lg - 1) >> 2);
        const int sh = isChroma ? lg - 2 : (lg + 1) >> 2;
        const uint32_t gmax = size == 4 ? 3u : size == 8 ? 5u : size == 16 ? 7u : 9u; /* lastSigXYGroupIndex[size - 1] */
#pragma unroll 1
        for (int c = 0; c < 2; c++) {
            uint8_t *mdl = M + (c ? CX_LASTY : CX_LASTX) + off;
            const uint32_t v = c ? ly : lx;
            const uint32_t g = v < 4 ? v : v < 6 ? 4u : v < 8 ? 5u : v < 12 ? 6u : v < 16 ? 7u : v < 24 ? 8u : 9u;
            uint32_t i = 0;
            for (; i < g; i++)
                bits += up_bin(&mdl[i >> sh], 1);
            if (g < gmax)
                bits += up_bin(&mdl[i >> sh], 0);
            if (g > 3)
                bits += ((g - 2) >> 1) * UP_ONE_BIT;
        }
    }
    const int scanPosLast = 16 * lastSet + (int)posLast;
    const uint32_t sigOff = isChroma ? 27 : 0;
    uint32_t ctxOff1 = 1, sbSigMemory = 0;
    int sbPrevDiag = -1;
#pragma unroll 1
    for (int sub = lastSet; sub >= 0; sub--) {
        int pattern = (int)(sbSigMemory & 3);
        const uint32_t smap = sigm[sub];
        if (sub != 0) {
            const int gy = c_rt.sb[lg - 2][sub] >> 4, gx = c_rt.sb[lg - 2][sub] & 15, diag = gy + gx;
            if (diag != sbPrevDiag) {
                sbSigMemory <<= 16;
                sbPrevDiag = diag;
            }
            if (sub != lastSet) {
                pattern = (int)((sbSigMemory >> (16 + gy)) & 3);
                const uint32_t flag = smap != 0;
                bits += up_bin(&M[CX_CG + (pattern != 0) + isChroma * 2], flag);
                if (!flag)
                    continue;
            }
            sbSigMemory += 1u << gy;
        }
        int nnz = 0;
        do { /* significance flags :3315 */
            int sigMap = (int)smap, pos, subPos = sub << 4, subPos2 = subPos;
            if (sub == lastSet) {
                absC[0] = (uint16_t)up_abs_at(p0, stride, lg, scan, sub, (int)posLast), nnz = 1;
                if (sigMap == 1)
                    break;
                pos = scanPosLast - 1;
                sigMap = (int)((uint32_t)sigMap << (31 - (pos & 15)));
            } else {
                if (sigMap == 1 && sub != 0) {
                    subPos2++;
                    absC[0] = (uint16_t)up_abs_at(p0, stride, lg, scan, sub, 0), nnz = 1;
                }
                pos = subPos + 15;
                sigMap = (int)((uint32_t)sigMap << 16);
            }
            uint32_t tOff;
            const uint8_t *map;
            if (lg == 2)
                tOff = 0, map = c_rt.ctx4[scan];
            else {
                tOff = lg == 3 ? (scan == 0 ? 9 : 15) : (!isChroma ? 21 : 12);
                tOff += (!isChroma && sub != 0) ? 3 : 0;
                map = c_rt.ctx8p[scan != 0][pattern];
            }
            do {
                const uint32_t f = sigMap < 0;
                const uint32_t ci = pos == 0 ? 0 : map[pos - subPos] + tOff;
                bits += up_bin(&M[CX_SIG + sigOff + ci], f);
                if (f) {
                    absC[nnz] = (uint16_t)up_abs_at(p0, stride, lg, scan, sub, pos - subPos);
                    nnz++;
                }
                sigMap = (int)((uint32_t)sigMap << 1);
                pos--;
            } while (pos >= subPos2);
        } while (0);
        /* levels :3386 */
        uint32_t rice = 0;
        uint32_t cset = (sub != 0 && !isChroma) ? 2 : 0;
        cset += ctxOff1 == 0;
        ctxOff1 = 1;
        const uint32_t o1 = isChroma * 16 + 4 * cset, o2 = isChroma * 4 + cset;
        const int nG1 = nnz < 8 ? nnz : 8;
        bits += UP_ONE_BIT * (uint32_t)nnz;
        int i = 0;
        for (; i < nG1; i++) {
            const int a = absC[i];
            bits += up_bin(&M[CX_G1 + o1 + ctxOff1], a > 1);
            if (a > 1) {
                bits += up_bin(&M[CX_G2 + o2], a > 2);
                if (a > 2) {
                    bits += up_golomb((uint32_t)a - 3, 0);
                    rice = a > 3;
                }
                i++;
                ctxOff1 = 0;
                break;
            }
            if (ctxOff1 < 3)
                ctxOff1++;
        }
        for (; i < nG1; i++) {
            const int a = absC[i];
            bits += up_bin(&M[CX_G1 + o1], a > 1);
            if (a > 1) {
                bits += up_golomb((uint32_t)a - 2, rice);
                if (rice < 4 && a > (int)(3u << rice))
                    rice++;
            }
        }
        for (; i < nnz; i++) {
            const int a = absC[i];
            bits += up_golomb((uint32_t)a - 1, rice);
            if (rice < 4 && a > (int)(3u << rice))
                rice++;
        }
    }
    return bits;
}
/* cooperative part: lanes `r` of `nl` build the significance maps of the unit's sub-blocks into sigm[] (LDS) */
__device__ __forceinline__ void rate_update_sigmaps(uint16_t *sigm, const int16_t *p0, uint32_t stride, int lg, const SvtAmdTuInfo ti, int r, int nl)
{
    const uint32_t scan = rate_scan_of(lg, ti);
    const int nsub = lg == 2 ? 1 : 1 << (2 * (lg - 2));
    for (int sb = r; sb < nsub; sb += nl)
        sigm[sb] = (uint16_t)up_sigmap(p0, stride, lg, scan, sb);
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
