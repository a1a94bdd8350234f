/*
 * oracle/svt_oracle_coeffscan.c - TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 * CPU restatement of the part of EncodeQuantizedCoefficients_generic (Codec/EbEntropyCoding.c:1172) that runs before the first bin of a transform
 * block is coded - scan choice :1347-1372, sub-block / scan re-ordering with significance maps :1378-1430, last significant position :1432-1461,
 * DC-only fast track :1308 - plus the sign / greater-1 patterns its level loop derives from the same values (:1532-1600), in the layout of
 * include/svt_hevc_amd.h (SvtAmdCoeffScanTu / Group / Lcu).  Pinned by tests/test_oracle_coeffscan.py: the reference's own function and a CABAC loop
 * fed with these records (integration/svt_coeff_scan_consumer.h, run on the reference's arithmetic coder) write the same bytes and leave the same
 * context models.  The scan tables are generated from their H.265 definitions (6.5.3), as in svt_oracle_rate.c.
 */
#include <string.h>
#include "svt_oracle.h"

static uint8_t g_diag4[16], g_col4[16], g_sb[4][64];
static int g_init;

static void init_tables(void)
{
    int n = 0;
    for (int d = 0; d < 7; d++)
        for (int y = d < 4 ? d : 3; y >= 0 && d - y < 4; y--)
            g_diag4[n++] = (uint8_t)(y * 4 + (d - y));
    for (int k = 0; k < 16; k++)
        g_col4[k] = (uint8_t)((k & 3) * 4 + (k >> 2));
    for (int lg = 0; lg < 4; lg++) {
        const int w = lg == 0 ? 2 : 1 << lg;
        n = 0;
        for (int d = 0; d < 2 * w - 1; d++)
            for (int y = d < w ? d : w - 1; y >= 0 && d - y < w; y--)
                g_sb[lg][n++] = (uint8_t)((y << 4) | (d - y));
    }
    g_init = 1;
}

static int ilog2(uint32_t v) { int n = 0; while (v > 1) v >>= 1, n++; return n; }

/* one transform block; groups / levels are appended at *ng / *nl (indices inside the LCU's lists) */
void svt_oracle_coeff_scan_tu(const int16_t *coeff, uint32_t stride, uint32_t size, uint32_t type, uint32_t intra_luma_mode, int is_chroma,
                              uint32_t nz, SvtAmdCoeffScanTu *tu, SvtAmdCoeffScanGroup *groups, uint32_t *ng, uint16_t *levels, uint32_t *nl)
{
    if (!g_init)
        init_tables();
    const int lg = ilog2(size);
    memset(tu, 0, sizeof(*tu));
    tu->first_group = (uint16_t)*ng;
    if (nz == 1 && coeff[0] != 0) { /* :1308 DC-only fast track: one coefficient at position 0 */
        SvtAmdCoeffScanGroup *g = &groups[(*ng)++];
        tu->dc_only = 1;
        g->sigmap = 1, g->sign = coeff[0] < 0, g->gt1 = (coeff[0] < 0 ? -coeff[0] : coeff[0]) > 1, g->first_level = (uint16_t)*nl;
        levels[(*nl)++] = (uint16_t)(coeff[0] < 0 ? -coeff[0] : coeff[0]);
        return;
    }
    uint32_t scan = 0; /* :1347 */
    if (type == 2 /* INTRA_MODE */ && lg <= 3 - is_chroma) {
        const int m = (int)intra_luma_mode; /* the chroma mode is EB_INTRA_CHROMA_DM (EncodeCoeff :4094): the luma mode decides for both */
        int d = 8 - ((m - 2) & 15);
        d = d < 0 ? -d : d;
        if (d <= 4)
            scan = (m & 16) ? 1 : 2;
    }
    tu->scan_index = (uint8_t)scan;
    const int nsub = 1 << (2 * (lg - 2));
    uint16_t sig[64];
    int16_t lin[64][16];
    int last = -1;
    for (int s = 0; s < nsub; s++) { /* :1378 */
        uint32_t gy = g_sb[lg - 2][s] >> 4, gx = g_sb[lg - 2][s] & 15;
        if (scan == 1) { const uint32_t t = gx; gx = gy, gy = t; }
        const int16_t *sb = coeff + 4 * gy * stride + 4 * gx;
        uint32_t m = 0;
        for (int k = 0; k < 16; k++) {
            const uint32_t pos = scan ? g_col4[k] : g_diag4[k]; /* scans4[scanIndex != SCAN_DIAG2] */
            uint32_t py = pos >> 2, px = pos & 3;
            if (scan == 1) { const uint32_t t = px; px = py, py = t; }
            lin[s][k] = sb[stride * py + px];
            m |= (uint32_t)(lin[s][k] != 0) << k;
        }
        sig[s] = (uint16_t)m;
        if (m)
            last = s;
    }
    tu->last_scan_set = (int8_t)last;
    if (last < 0)
        return;
    const uint32_t pos_last = (uint32_t)ilog2(sig[last]); /* :1444 */
    uint32_t ly = 4 * (uint32_t)(g_sb[lg - 2][last] >> 4), lx = 4 * (uint32_t)(g_sb[lg - 2][last] & 15);
    const uint32_t pl = scan ? g_col4[pos_last] : g_diag4[pos_last];
    ly += pl >> 2, lx += pl & 3;
    if (scan) { const uint32_t t = lx; lx = ly, ly = t; }
    tu->pos_last = (uint8_t)pos_last, tu->last_x = (uint8_t)lx, tu->last_y = (uint8_t)ly;
    for (int s = last; s >= 0; s--) { /* the level loop's view of a sub-block (:1532-1600): coded coefficients = set bits from the top */
        SvtAmdCoeffScanGroup *g = &groups[(*ng)++];
        g->sigmap = sig[s], g->sign = 0, g->gt1 = 0, g->first_level = (uint16_t)*nl;
        int i = 0;
        for (int k = 15; k >= 0; k--)
            if (sig[s] >> k & 1) {
                const int v = lin[s][k], a = v < 0 ? -v : v;
                g->sign = (uint16_t)(g->sign * 2 + (v < 0));
                g->gt1 |= (uint16_t)((a > 1) << i);
                levels[(*nl)++] = (uint16_t)a;
                i++;
            }
    }
}

/* every transform block of an LCU's encode-pass records (the 8- and 16-bit records share their heads); group_base / level_base are left 0.
 * returns the number of blocks with coefficients */
int svt_oracle_coeff_scan_lcu(const SvtAmdLcuWork *W, const SvtAmdLcuResult *R, SvtAmdCoeffScanLcu *out, SvtAmdCoeffScanGroup *groups, uint16_t *levels)
{
    uint32_t ng = 0, nl = 0;
    int coded = 0;
    memset(out, 0, sizeof(*out));
    for (int p = 0; p < 3; p++)
        for (int c = 0; c < SVT_AMD_LCU_MAX_CUS; c++)
            out->tu[p][c].last_scan_set = -1;
    const int big = W->num_cus == 1 && W->cu[0].size == 64;
    const int slots = big ? 5 : W->num_cus;
    for (int c = big ? 1 : 0; c < slots; c++) {
        const SvtAmdLcuCu *u = &W->cu[big ? 0 : c];
        const uint32_t size = big ? 32 : u->size, x = big ? 32u * ((c - 1) & 1) : u->x, y = big ? 32u * ((c - 1) >> 1) : u->y;
        for (int p = 0; p < 3; p++) {
            SvtAmdCoeffScanTu *tu = &out->tu[p][c];
            if (!R->cu[c].cbf[p])
                continue;
            const uint32_t ts = p ? (size == 8 ? 4 : size >> 1) : size;
            const int16_t *co = p == 0 ? R->coeff_y + 64 * y + x : (p == 1 ? R->coeff_cb : R->coeff_cr) + 32 * (y >> 1) + (x >> 1);
            svt_oracle_coeff_scan_tu(co, p ? 32 : 64, ts, u->pred_mode, u->intra_luma_mode, p != 0, R->cu[c].nz[p], tu, groups, &ng, levels, &nl);
            coded += tu->last_scan_set >= 0;
        }
    }
    out->groups = (uint16_t)ng, out->levels = (uint16_t)nl;
    return coded;
}
