/*
 * oracle/svt_oracle_rate.c - TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 * CPU restatement of the coefficient-rate estimator the reference's mode decision and encode pass use:
 *   EstimateQuantizedCoefficients_Lossy      Codec/EbCoeffEstimation_Intrinsic.c:1415-1842
 *   EncodeLastSignificantXYTemp :44-70, RemainingCoeffExponentialGolombCodeTemp :309-326
 * (slot [1][*] of the table EstimateQuantizedCoefficients, Codec/EbEntropyCoding.h:334-347; callers
 * TuEstimateCoeffBits* Codec/EbEntropyCoding.c:7832-8100, DecoupledQuantizeInvQuantizeLoops EbTransforms.c:2717).
 * It prices a quantised TU from the precomputed CabacCost_t tables: last position, coded_sub_block flags,
 * significance flags (H.265 9.3.4.2.5 context maps for prevCsbf = 0 only - the "lossy" simplification),
 * greater-1/greater-2 flags with a fixed context set, Exp-Golomb remainders with rice parameter 0, sign bits.
 * The scan / context tables are generated from their H.265 definitions (6.5.3 up-right diagonal scan,
 * 9.3.4.2.5 ctxIdxMap); pinned by tests/test_oracle_rate.py against the reference symbol.
 */
#include <string.h>
#include "svt_oracle.h"

#define ONE_BIT 32

static uint8_t g_diag4[16], g_col4[16], g_sb[4][64], g_ctx4[3][16], g_ctx8[2][16];
static int g_init;

static void init_tables(void)
{
    static const uint8_t ctxIdxMap[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8}; /* H.265 Table 9-41, 4x4 raster */
    int n = 0;
    for (int d = 0; d < 7; d++) /* up-right diagonal scan of a 4x4 block, position = y*4 + x */
        for (int y = d < 4 ? d : 3; y >= 0 && d - y < 4; y--)
            g_diag4[n++] = (uint8_t)(y * 4 + (d - y));
    for (int k = 0; k < 16; k++)
        g_col4[k] = (uint8_t)((k & 3) * 4 + (k >> 2));
    for (int lg = 0; lg < 4; lg++) { /* sub-block scans of 4x4 (1), 8x8 (2x2), 16x16 (4x4), 32x32 (8x8); (y << 4) | x */
        const int w = lg == 0 ? 2 : 1 << lg; /* the reference uses the 8x8 table for 4x4 too (only entry 0 is read) */
        n = 0;
        for (int d = 0; d < 2 * w - 1; d++)
            for (int y = d < w ? d : w - 1; y >= 0 && d - y < w; y--)
                g_sb[lg][n++] = (uint8_t)((y << 4) | (d - y));
    }
    for (int k = 0; k < 16; k++) {
        g_ctx4[0][k] = ctxIdxMap[g_diag4[k]];
        g_ctx4[1][k] = ctxIdxMap[k];
        g_ctx4[2][k] = ctxIdxMap[g_col4[k]];
        const int sd = (g_diag4[k] >> 2) + (g_diag4[k] & 3), sr = (k >> 2) + (k & 3);
        g_ctx8[0][k] = sd == 0 ? 2 : sd < 3 ? 1 : 0; /* sigCtx for prevCsbf == 0, in diagonal scan order */
        g_ctx8[1][k] = sr == 0 ? 2 : sr < 3 ? 1 : 0; /* same, raster order */
    }
    g_init = 1;
}

static inline uint32_t ilog2u(uint32_t v) { uint32_t n = 0; while (v > 1) v >>= 1, n++; return n; }

static uint32_t golomb_bits(uint32_t symbol, uint32_t param) /* :309 */
{
    int32_t cw = (int32_t)(symbol >> param);
    uint32_t bins = param + 1;
    if (cw < 3)
        bins += (uint32_t)cw;
    else {
        cw -= 2;
        bins += 2 * ilog2u((uint32_t)cw) + 3;
    }
    return ONE_BIT * bins;
}

static uint32_t last_xy_bits(const SvtAmdCabacCost *C, uint32_t x, uint32_t y, uint32_t size, int isChroma) /* :44 */
{
    const int32_t off = (isChroma ? 120 : 0) - 8;
    if (size == 1)
        return C->CabacBitsLast[0] + C->CabacBitsLast[1];
    return C->CabacBitsLast[off + 2 * (int32_t)(x + size) + 0] + C->CabacBitsLast[off + 2 * (int32_t)(y + size) + 1];
}

/* returns the amount the reference adds to *coeffBitsLong; numNonZeroCoeffs must be the true count (>= 1) */
uint64_t svt_oracle_coeff_bits_lossy(const SvtAmdCabacCost *C, uint32_t size, uint32_t type, uint32_t intraLumaMode,
                                     uint32_t intraChromaMode, const int16_t *coeff, uint32_t stride,
                                     uint32_t componentType, uint32_t numNonZeroCoeffs)
{
    if (!g_init)
        init_tables();
    const int isChroma = componentType != 0;
    const uint32_t lg = ilog2u(size);
    uint32_t bits = 0, scan = 0;
    uint16_t lin[32 * 32], sigmaps[64], g1maps[64];

    if (numNonZeroCoeffs == 1 && coeff[0] != 0) { /* DC-only fast track :1452 */
        const int a = coeff[0] < 0 ? -coeff[0] : coeff[0];
        const uint32_t o1 = isChroma * 16, o2 = isChroma * 4;
        bits += last_xy_bits(C, 0, 0, size, isChroma);
        bits += C->CabacBitsG1[2 * (o1 + 1) + (a > 1)];
        if (a > 1) {
            bits += C->CabacBitsG2[2 * o2 + (a > 2)];
            if (a > 2)
                bits += golomb_bits((uint32_t)a - 3, 0);
        }
        bits += ONE_BIT;
        return (uint64_t)bits << 10;
    }
    if (type == 2 /* INTRA_MODE */ && lg <= (uint32_t)(3 - isChroma)) { /* mode-dependent scan :1490 */
        static const uint32_t chromaMap[5] = {0, 26, 10, 1, 4};
        const uint32_t tc = chromaMap[intraChromaMode];
        const int32_t m = (!isChroma || tc == 4) ? (int32_t)intraLumaMode : (int32_t)tc;
        const int32_t dlt = 8 - ((m - 2) & 15);
        if ((dlt < 0 ? -dlt : dlt) <= 4)
            scan = (m & 16) ? 1 : 2; /* SCAN_HOR2 : SCAN_VER2 */
    }
    int32_t lastSet = -1, sub = 0;
    for (;; sub++) { /* scan-order linearisation + per-subblock maps :1515 */
        uint32_t gy = g_sb[lg - 2][sub] >> 4, gx = g_sb[lg - 2][sub] & 15;
        if (scan == 1) { const uint32_t tmp = gx; gx = gy; gy = tmp; }
        const int16_t *p = coeff + 4 * gy * stride + 4 * gx;
        uint32_t sig = 0, g1 = 0, num = 0;
        for (int k = 0; k < 16; k++) {
            const uint32_t pos = scan ? g_col4[k] : g_diag4[k];
            uint32_t py = pos >> 2, px = pos & 3;
            if (scan == 1) { const uint32_t tmp = px; px = py; py = tmp; }
            const int v = p[stride * py + px], a = v < 0 ? -v : v;
            lin[16 * sub + k] = (uint16_t)a;
            num += a != 0;
            sig |= (uint32_t)(a != 0) << k;
            g1 |= (uint32_t)(a > 1) << k;
        }
        sigmaps[sub] = (uint16_t)sig, g1maps[sub] = (uint16_t)g1;
        if (sig) {
            lastSet = sub;
            numNonZeroCoeffs -= num;
            if (numNonZeroCoeffs == 0)
                break;
        }
    }
    /* last significant position :1589 */
    const uint32_t posLast = ilog2u(sigmaps[lastSet]);
    uint32_t ly = 4 * (g_sb[lg - 2][lastSet] >> 4), lx = 4 * (g_sb[lg - 2][lastSet] & 15);
    const uint32_t pl = scan ? g_col4[posLast] : g_diag4[posLast];
    ly += pl >> 2, lx += pl & 3;
    const int32_t scanPosLast = 16 * lastSet + (int32_t)posLast;
    if (scan) { const uint32_t tmp = lx; lx = ly; ly = tmp; }
    bits += last_xy_bits(C, lx, ly, size, isChroma);

    const uint32_t sigOff = isChroma ? 27 : 0;
    for (sub = lastSet; sub >= 0; sub--) { /* :1624 */
        int32_t nnz = 0, absC[16] = {0};
        if (sub != 0 && sub != lastSet) {
            const uint32_t flag = sigmaps[sub] != 0;
            bits += C->CabacBitsSigMl[2 * (isChroma * 2) + flag];
            if (!flag)
                continue;
        }
        do { /* significance flags */
            int32_t sigMap = sigmaps[sub], pos, subPos = sub << 4, subPos2 = subPos;
            if (sub == lastSet) {
                absC[0] = lin[scanPosLast], nnz = 1;
                if (sigMap == 1)
                    break;
                pos = scanPosLast - 1;
                sigMap = (int32_t)((uint32_t)sigMap << (31 - (pos & 15)));
            } else {
                if (sigMap == 1 && sub != 0) {
                    subPos2++;
                    absC[0] = lin[subPos], nnz = 1;
                }
                pos = subPos + 15;
                sigMap = (int32_t)((uint32_t)sigMap << 16);
            }
            if (sub == 0)
                subPos2 = 1;
            uint32_t tOff;
            const uint8_t *map;
            if (lg == 2)
                tOff = 0, map = g_ctx4[scan];
            else {
                tOff = lg == 3 ? (scan == 0 ? 9 : 15) : (!isChroma ? 21 : 12);
                tOff += (!isChroma && sub != 0) ? 3 : 0;
                map = g_ctx8[scan != 0] - subPos;
            }
            const uint8_t *bp = C->CabacBitsSig + 2 * sigOff + 2 * tOff;
            while (pos >= subPos2) {
                const int f = sigMap < 0;
                bits += bp[2 * map[pos] + f];
                if (f)
                    absC[nnz++] = lin[pos];
                sigMap = (int32_t)((uint32_t)sigMap << 1);
                pos--;
            }
            if (pos == 0) {
                const int f = sigMap < 0;
                bits += C->CabacBitsSig[2 * sigOff + f];
                if (f)
                    absC[nnz++] = lin[pos];
            }
        } while (0);
        /* level values :1757 */
        const uint32_t cset = (sub != 0 && !isChroma) ? 2 : 0;
        const uint32_t o1 = isChroma * 16 + 4 * cset, o2 = isChroma * 4 + cset;
        const int32_t nG1 = nnz < 8 ? nnz : 8;
        bits += ONE_BIT * (uint32_t)nnz;
        if (g1maps[sub] == 0) {
            if (nnz > 0)
                bits += C->CabacBitsG1x[4 * o1 + nnz - 1];
            continue;
        }
        int32_t i = 0;
        for (; i < nG1; i++) {
            const int32_t a = absC[i];
            bits += C->CabacBitsG1[2 * (o1 + 1) + (a > 1)];
            if (a > 1) {
                bits += C->CabacBitsG2[2 * o2 + (a > 2)];
                if (a > 2)
                    bits += golomb_bits((uint32_t)a - 3, 0);
                i++;
                break;
            }
        }
        for (; i < nG1; i++) {
            const int32_t a = absC[i];
            bits += C->CabacBitsG1[2 * o1 + (a > 1)];
            if (a > 1)
                bits += golomb_bits((uint32_t)a - 2, 0);
        }
        for (; i < nnz; i++)
            bits += golomb_bits((uint32_t)absC[i] - 1, 0);
    }
    return (uint64_t)bits << 10;
}
