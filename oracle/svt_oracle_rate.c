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

/* =====================================================================================================================
 * The CABAC-context-UPDATING estimator: EstimateQuantizedCoefficients_generic_Update (Codec/EbEntropyCoding.c:2986-3480)
 * with EstimateLastSignificantXY_UPDATE (:2374-2461) and EstimateRemainingCoeffExponentialGolombCode
 * (Codec/EbEntropyCodingUtil.c:326-343).  Slot [0] of the table EstimateQuantizedCoefficientsUpdate
 * (Codec/EbEntropyCoding.h:387-392); reached from TuEstimateCoeffBitsLuma / TuEstimateCoeffBits_R / ..EncDec
 * (EbEntropyCoding.c:7900-8100) when the mode decision runs with coeffCabacUpdate (EbEncDecProcess.c:2115-2123).
 * Every context-coded bin costs CabacEstimatedBits[bin ^ state] and moves the state of ITS context model
 * (UPDATE_CONTEXT_MODEL, Codec/EbMdRateEstimation.h:118) in the caller's CoeffCtxtMdl_t, which the caller threads from
 * candidate to candidate and from coding unit to coding unit (EbProductCodingLoop.c:4405, 5053).
 *
 * The model is handed over as SVT_ORACLE_COEFF_CTX_WORDS uint32 words in CoeffCtxtMdl_t's order
 * (Codec/EbCabacContextModel.h:204-214): lastSigX[30] lastSigY[30] sig[42] coeffGroupSig[4] greater1[24] greater2[6].
 * State word = (pStateIdx << 1) | valMps.  The transition table is generated from H.265 Table 9-41 (transIdxLps; the MPS
 * path is min(pStateIdx + 1, 62)); the 128 entropy-bit constants are HM's ContextModel::m_entropyBits as the reference
 * carries them (Codec/EbHmCode.c:236-246) - data, no formula reproduces them.  Both are pinned against the reference's
 * exported arrays by tests/test_oracle_rate.py.
 * ===================================================================================================================== */
#define UP_ONE_BIT 32768u
enum { CX_LASTX = 0, CX_LASTY = 30, CX_SIG = 60, CX_CG = 102, CX_G1 = 106, CX_G2 = 130 };

const uint32_t svt_oracle_cabac_estimated_bits[128] = {
    0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
    0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
    0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
    0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
    0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
    0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
    0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
    0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb};
uint32_t svt_oracle_next_state_mps_lps[256];
static uint8_t g_ctx8p[2][4][16]; /* sigCtx by (scan != diagonal, prevCsbf pattern, position in scan order): 9.3.4.2.5 */
static int g_up_init;

static void init_update_tables(void)
{
    static const uint8_t transIdxLps[64] = {0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9,  11, 11, 12, 13, 13, 15, 15, 16, 16,
                                            18, 18, 19, 19, 21, 21, 22, 22, 23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30,
                                            31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63};
    if (!g_init)
        init_tables();
    for (uint32_t s = 0; s < 128; s++) {
        const uint32_t p = s >> 1, mps = s & 1;
        svt_oracle_next_state_mps_lps[s] = ((p < 62 ? p + 1 : p) << 1) | mps;                 /* bin == valMps */
        svt_oracle_next_state_mps_lps[128 + s] = p == 0 ? (mps ^ 1) : ((uint32_t)transIdxLps[p] << 1) | mps;
    }
    for (int m = 0; m < 2; m++)
        for (int pat = 0; pat < 4; pat++)
            for (int k = 0; k < 16; k++) {
                const int pos = m ? g_col4[k] : g_diag4[k], yP = pos >> 2, xP = pos & 3;
                g_ctx8p[m][pat][k] = pat == 0 ? (xP + yP == 0 ? 2 : xP + yP < 3 ? 1 : 0)
                                     : pat == 1 ? (yP == 0 ? 2 : yP == 1 ? 1 : 0)
                                     : pat == 2 ? (xP == 0 ? 2 : xP == 1 ? 1 : 0) : 2;
            }
    g_up_init = 1;
}

static inline uint32_t bin_cost(uint32_t *m, uint32_t bin) /* cost of one context-coded bin + state transition */
{
    const uint32_t st = *m, c = svt_oracle_cabac_estimated_bits[bin ^ st];
    *m = svt_oracle_next_state_mps_lps[((bin ^ (st & 1)) << 7) | st];
    return c;
}

static uint32_t golomb_bits_up(uint32_t symbol, uint32_t param) /* EbEntropyCodingUtil.c:326 */
{
    int32_t cw = (int32_t)(symbol >> param);
    uint32_t bins = param + 1;
    if (cw < 3)
        bins += (uint32_t)cw;
    else {
        cw -= 2;
        bins += 2 * ilog2u((uint32_t)cw) + 3;
    }
    return UP_ONE_BIT * bins;
}

static uint32_t last_xy_update(uint32_t *M, uint32_t x, uint32_t y, uint32_t size, uint32_t lg, int isChroma) /* :2374 */
{
    static const uint8_t grp[32] = {0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9};
    const int32_t off = isChroma ? 15 : (int32_t)((lg - 2) * 3 + ((lg - 1) >> 2));
    const int32_t sh = isChroma ? (int32_t)lg - 2 : (int32_t)((lg + 1) >> 2);
    uint32_t bits = 0;
    for (int c = 0; c < 2; c++) {
        uint32_t *mdl = M + (c ? CX_LASTY : CX_LASTX);
        const uint32_t g = grp[c ? y : x];
        uint32_t i = 0;
        for (; i < g; i++)
            bits += bin_cost(&mdl[off + (int32_t)(i >> sh)], 1);
        if (g < grp[size - 1])
            bits += bin_cost(&mdl[off + (int32_t)(i >> sh)], 0);
        if (g > 3)
            bits += ((g - 2) >> 1) * UP_ONE_BIT;
    }
    return bits;
}

/* returns what the reference adds to *coeffBitsLong and leaves the updated states in ctx[] */
uint64_t svt_oracle_coeff_bits_update(uint32_t *ctx, uint32_t size, uint32_t type, uint32_t intraLumaMode,
                                      uint32_t intraChromaMode, const int16_t *coeff, uint32_t stride,
                                      uint32_t componentType, uint32_t numNonZeroCoeffs)
{
    if (!g_up_init)
        init_update_tables();
    const int isChroma = componentType != 0;
    const uint32_t lg = ilog2u(size);
    uint32_t bits = 0, scan = 0;
    uint16_t lin[32 * 32], sigmaps[64];

    if (numNonZeroCoeffs == 1 && coeff[0] != 0) { /* DC-only fast track :3064 */
        const int32_t off = isChroma ? 15 : (int32_t)((lg - 2) * 3 + ((lg - 1) >> 2));
        const int a = coeff[0] < 0 ? -coeff[0] : coeff[0];
        bits += bin_cost(&ctx[CX_LASTX + off], 0);
        bits += bin_cost(&ctx[CX_LASTY + off], 0);
        bits += bin_cost(&ctx[CX_G1 + isChroma * 16 + 1], a > 1);
        if (a > 1) {
            bits += bin_cost(&ctx[CX_G2 + isChroma * 4], a > 2);
            if (a > 2)
                bits += golomb_bits_up((uint32_t)a - 3, 0);
        }
        bits += UP_ONE_BIT;
        return bits;
    }
    if (type == 2 /* INTRA_MODE */ && lg <= (uint32_t)(3 - isChroma)) { /* mode-dependent scan :3132 */
        static const uint32_t chromaMap[5] = {0, 26, 10, 1, 4};
        const uint32_t tc = chromaMap[intraChromaMode];
        const int32_t m = (!isChroma || tc == 4) ? (int32_t)intraLumaMode : (int32_t)tc;
        const int32_t dlt = 8 - ((m - 2) & 15);
        if ((dlt < 0 ? -dlt : dlt) <= 4)
            scan = (m & 16) ? 1 : 2; /* SCAN_HOR2 : SCAN_VER2 */
    }
    int32_t lastSet = -1, sub = 0;
    for (;; sub++) { /* :3165 */
        uint32_t gy = g_sb[lg - 2][sub] >> 4, gx = g_sb[lg - 2][sub] & 15;
        if (scan == 1) { const uint32_t tmp = gx; gx = gy; gy = tmp; }
        const int16_t *p = coeff + 4 * gy * stride + 4 * gx;
        uint32_t sig = 0, num = 0;
        for (int k = 0; k < 16; k++) {
            const uint32_t pos = scan ? g_col4[k] : g_diag4[k];
            uint32_t py = pos >> 2, px = pos & 3;
            if (scan == 1) { const uint32_t tmp = px; px = py; py = tmp; }
            const int v = p[stride * py + px], a = v < 0 ? -v : v;
            lin[16 * sub + k] = (uint16_t)a;
            num += a != 0;
            sig |= (uint32_t)(a != 0) << k;
        }
        sigmaps[sub] = (uint16_t)sig;
        if (sig) {
            lastSet = sub;
            numNonZeroCoeffs -= num;
            if (numNonZeroCoeffs == 0)
                break;
        }
    }
    const uint32_t posLast = ilog2u(sigmaps[lastSet]); /* :3218 */
    uint32_t ly = 4 * (g_sb[lg - 2][lastSet] >> 4), lx = 4 * (g_sb[lg - 2][lastSet] & 15);
    const uint32_t pl = scan ? g_col4[posLast] : g_diag4[posLast];
    ly += pl >> 2, lx += pl & 3;
    const int32_t scanPosLast = 16 * lastSet + (int32_t)posLast;
    if (scan) { const uint32_t tmp = lx; lx = ly; ly = tmp; }
    bits += last_xy_update(ctx, lx, ly, size, lg, isChroma);

    const uint32_t sigOff = isChroma ? 27 : 0;
    uint32_t ctxOff1 = 1, sbSigMemory = 0;
    int32_t sbPrevDiag = -1;
    for (sub = lastSet; sub >= 0; sub--) { /* :3266 */
        int32_t pattern = (int32_t)(sbSigMemory & 3);
        if (sub != 0) {
            const int32_t gy = g_sb[lg - 2][sub] >> 4, gx = g_sb[lg - 2][sub] & 15, diag = gy + gx;
            if (diag != sbPrevDiag) {
                sbSigMemory <<= 16;
                sbPrevDiag = diag;
            }
            if (sub != lastSet) {
                pattern = (int32_t)((sbSigMemory >> (16 + gy)) & 3);
                const uint32_t flag = sigmaps[sub] != 0;
                bits += bin_cost(&ctx[CX_CG + (pattern != 0) + isChroma * 2], flag);
                if (!flag)
                    continue;
            }
            sbSigMemory += 1u << gy;
        }
        int32_t nnz = 0, absC[16] = {0};
        do { /* significance flags :3315 */
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
            uint32_t tOff;
            const uint8_t *map;
            if (lg == 2)
                tOff = 0, map = g_ctx4[scan];
            else {
                tOff = lg == 3 ? (scan == 0 ? 9 : 15) : (!isChroma ? 21 : 12);
                tOff += (!isChroma && sub != 0) ? 3 : 0;
                map = g_ctx8p[scan != 0][pattern] - subPos;
            }
            do {
                const uint32_t f = sigMap < 0;
                const uint32_t ci = pos == 0 ? 0 : map[pos] + tOff;
                bits += bin_cost(&ctx[CX_SIG + sigOff + ci], f);
                if (f) {
                    absC[nnz] = pos >= 0 ? lin[pos] : 0;
                    nnz++;
                }
                sigMap = (int32_t)((uint32_t)sigMap << 1);
                pos--;
            } while (pos >= subPos2);
        } while (0);
        /* levels :3386 */
        uint32_t rice = 0;
        uint32_t cset = (sub != 0 && !isChroma) ? 2 : 0;
        cset += ctxOff1 == 0;
        ctxOff1 = 1;
        const uint32_t o1 = isChroma * 16 + 4 * cset, o2 = isChroma * 4 + cset;
        const int32_t nG1 = nnz < 8 ? nnz : 8;
        bits += UP_ONE_BIT * (uint32_t)nnz;
        int32_t i = 0;
        for (; i < nG1; i++) {
            const int32_t a = absC[i];
            bits += bin_cost(&ctx[CX_G1 + o1 + ctxOff1], a > 1);
            if (a > 1) {
                bits += bin_cost(&ctx[CX_G2 + o2], a > 2);
                if (a > 2) {
                    bits += golomb_bits_up((uint32_t)a - 3, 0);
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
            const int32_t a = absC[i];
            bits += bin_cost(&ctx[CX_G1 + o1], a > 1);
            if (a > 1) {
                bits += golomb_bits_up((uint32_t)a - 2, rice);
                if (rice < 4 && a > (int32_t)(3u << rice))
                    rice++;
            }
        }
        for (; i < nnz; i++) {
            const int32_t a = absC[i];
            bits += golomb_bits_up((uint32_t)a - 1, rice);
            if (rice < 4 && a > (int32_t)(3u << rice))
                rice++;
        }
    }
    return bits;
}
