/*
 * oracle/svt_oracle_intra.c - TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 * CPU restatement of the 24 intra-prediction leaf kernels (8- and 16-bit) of
 * /root/reference/Source/Lib/C_DEFAULT/EbIntraPrediction_C.c:15-1051.
 *
 * Reference-sample layout (size N): ref[0..2N-1] left column top-to-bottom then
 * bottom-left, ref[2N] top-left, ref[2N+1..4N] top row then top-right.
 * `skip` = 1 predicts even rows only.  mode ids: SVT_ORACLE_INTRA_*.
 */
#include <stdlib.h>
#include "svt_oracle.h"

#define GET(p, i) (bps == 1 ? (int)((const uint8_t *)(p))[i] : (int)((const uint16_t *)(p))[i])
#define PUT(p, i, v)                                   \
    do {                                               \
        if (bps == 1)                                  \
            ((uint8_t *)(p))[i] = (uint8_t)(v);        \
        else                                           \
            ((uint16_t *)(p))[i] = (uint16_t)(v);      \
    } while (0)

static inline int clipv(int v, int maxv) { return v < 0 ? 0 : v > maxv ? maxv : v; }
static inline int ilog2(uint32_t v) { int n = 0; while (v > 1) v >>= 1, n++; return n; }

void svt_oracle_IntraPred(int mode, int bps, uint32_t size, const void *ref, void *pred, uint32_t stride, int skip,
                          int32_t intraPredAngle)
{
    const int maxv = bps == 1 ? 255 : 1023; /* MAX_SAMPLE_VALUE / MAX_SAMPLE_VALUE_10BIT */
    const uint32_t N = size, rs = skip ? 2 : 1, L = 0, TL = 2 * N, T = 2 * N + 1;
    uint32_t x, y;
    switch (mode) {
    case SVT_ORACLE_INTRA_VERTICAL_LUMA:   /* :15-60 */
    case SVT_ORACLE_INTRA_VERTICAL_CHROMA: /* :111-141 */
        for (y = 0; y < N; y += rs)
            for (x = 0; x < N; x++)
                PUT(pred, y * stride + x, GET(ref, T + x));
        if (mode == SVT_ORACLE_INTRA_VERTICAL_LUMA && N < 32)
            for (y = 0; y < N; y += rs)
                PUT(pred, y * stride, clipv(GET(pred, y * stride) + ((GET(ref, L + y) - GET(ref, TL)) >> 1), maxv));
        break;
    case SVT_ORACLE_INTRA_HORIZONTAL_LUMA:   /* :174-220 */
    case SVT_ORACLE_INTRA_HORIZONTAL_CHROMA: /* :270-301 */
        for (y = 0; y < N; y += rs)
            for (x = 0; x < N; x++)
                PUT(pred, y * stride + x, GET(ref, L + y));
        if (mode == SVT_ORACLE_INTRA_HORIZONTAL_LUMA && N < 32)
            for (x = 0; x < N; x++)
                PUT(pred, x, clipv(GET(pred, x) + ((GET(ref, TL + x + 1) - GET(ref, TL)) >> 1), maxv));
        break;
    case SVT_ORACLE_INTRA_DC_LUMA:   /* :337-406 */
    case SVT_ORACLE_INTRA_DC_CHROMA: /* :480-533 */ {
        uint32_t sum = 0;
        for (x = 0; x < N; x++)
            sum += (uint32_t)GET(ref, T + x) + (uint32_t)GET(ref, L + x);
        const int dc = bps == 1 ? (uint8_t)((sum + N) >> ilog2(N << 1)) : (uint16_t)((sum + N) >> ilog2(N << 1));
        for (y = 0; y < N; y += rs)
            for (x = 0; x < N; x++)
                PUT(pred, y * stride + x, dc);
        if (mode == SVT_ORACLE_INTRA_DC_LUMA && N < 32) {
            PUT(pred, 0, (GET(ref, L) + GET(ref, T) + (GET(pred, 0) << 1) + 2) >> 2);
            for (x = 1; x < N; x++)
                PUT(pred, x, (GET(ref, T + x) + 3 * GET(pred, x) + 2) >> 2);
            for (y = rs; y < N; y += rs)
                PUT(pred, y * stride, (GET(ref, L + y) + 3 * GET(pred, y * stride) + 2) >> 2);
        }
        break;
    }
    case SVT_ORACLE_INTRA_PLANAR: /* :591-636 */ {
        const int tr = GET(ref, T + N), bl = GET(ref, L + N), shift = ilog2(N) + 1;
        for (y = 0; y < N; y += rs)
            for (x = 0; x < N; x++)
                PUT(pred, y * stride + x,
                    (int)(((N - 1 - x) * (uint32_t)GET(ref, L + y) + (x + 1) * (uint32_t)tr +
                           (N - 1 - y) * (uint32_t)GET(ref, T + x) + (y + 1) * (uint32_t)bl + N) >> shift));
        break;
    }
    case SVT_ORACLE_INTRA_ANGULAR_34: /* :685-716 */
        for (y = 0; y < N; y += rs)
            for (x = 0; x < N; x++)
                PUT(pred, y * stride + x, GET(ref, T + y + x + 1));
        break;
    case SVT_ORACLE_INTRA_ANGULAR_18: /* :752-788 */
        for (y = 0; y < N; y += rs)
            for (x = 0; x < N; x++)
                PUT(pred, y * stride + x, GET(ref, TL - y + x));
        break;
    case SVT_ORACLE_INTRA_ANGULAR_2: /* :829-857 */
        for (y = 0; y < N; y += rs)
            for (x = 0; x < N; x++)
                PUT(pred, y * stride + x, GET(ref, L + y + x + 1));
        break;
    case SVT_ORACLE_INTRA_ANGULAR_VERTICAL: /* :891-930, ref = refSampMain */ {
        int32_t deltaSum = intraPredAngle;
        for (y = 0; y < N; y += rs) {
            const int32_t di = deltaSum >> 5;
            const int32_t df = deltaSum & 31;
            for (x = 0; x < N; x++)
                PUT(pred, y * stride + x,
                    ((32 - df) * GET(ref, 1 + di + (int32_t)x) + df * GET(ref, 1 + di + (int32_t)x + 1) + 16) >> 5);
            deltaSum += (int32_t)rs * intraPredAngle;
        }
        break;
    }
    case SVT_ORACLE_INTRA_ANGULAR_HORIZONTAL: /* :974-1013, ref = refSampMain */ {
        int32_t deltaSum = 0;
        for (x = 0; x < N; x++) {
            deltaSum += intraPredAngle;
            const int32_t di = deltaSum >> 5;
            const int32_t df = deltaSum & 31;
            for (y = 0; y < N; y += rs)
                PUT(pred, y * stride + x,
                    ((32 - df) * GET(ref, 1 + di + (int32_t)y) + df * GET(ref, 1 + di + (int32_t)y + 1) + 16) >> 5);
        }
        break;
    }
    default:
        break;
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Encode-pass intra prediction of one prediction unit from neighbour-array slices:
 *   GenerateIntraReferenceSamplesEncodePass   Codec/EbIntraPrediction.c:212-757  (16-bit twin :760-1300)
 *   EncodePassIntraPrediction                 Codec/EbIntraPrediction.c:4395-4673 (16-bit :4680-4960)
 * restated sample-wise in the H.265 8.4.4.2 form (the reference's kernels are the same arithmetic specialised per mode;
 * C_DEFAULT/EbIntraPrediction_C.c).  Pinned by tests/test_oracle_intra_golden.py on records of real call pairs.
 * ------------------------------------------------------------------------------------------------------------------ */
static int pu_predict(int mode, int N, const int *r, int x, int y, int dc, int lumaEdge, int maxv)
{
    static const int ang[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32}, inv[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256};
    const int *left = r, *top = r + 2 * N + 1, tl = r[2 * N], lg = ilog2((uint32_t)N);
    if (mode == 0) /* planar */
        return ((N - 1 - x) * left[y] + (x + 1) * top[N] + (N - 1 - y) * top[x] + (y + 1) * left[N] + N) >> (lg + 1);
    if (mode == 1) { /* DC */
        if (lumaEdge && N < 32) {
            if (x == 0 && y == 0)
                return (left[0] + top[0] + 2 * dc + 2) >> 2;
            if (y == 0)
                return (top[x] + 3 * dc + 2) >> 2;
            if (x == 0)
                return (left[y] + 3 * dc + 2) >> 2;
        }
        return dc;
    }
    if (mode == 26)
        return (lumaEdge && N < 32 && x == 0) ? clipv(top[0] + ((left[y] - tl) >> 1), maxv) : top[x];
    if (mode == 10)
        return (lumaEdge && N < 32 && y == 0) ? clipv(left[0] + ((top[x] - tl) >> 1), maxv) : left[y];
    const int vert = mode >= 18, d = vert ? mode - 26 : 10 - mode, a = d < 0 ? -ang[-d] : ang[d];
    const int u = vert ? x : y, v = vert ? y : x, *mainr = vert ? top : left, *side = vert ? left : top;
    const int pos = (v + 1) * a, i = pos >> 5, f = pos & 31;
    int s[2];
    for (int k = 0; k < 2; k++) {
        const int idx = u + i + 1 + k; /* main[0] = top-left, main[j] = mainr[j-1], negative: projected side sample */
        s[k] = idx > 0 ? mainr[idx - 1] : idx == 0 ? tl : side[((-idx * inv[-d] + 128) >> 8) - 1];
    }
    return ((32 - f) * s[0] + f * s[1] + 16) >> 5;
}

void svt_oracle_intra_pu(int bps, const SvtAmdIntraPuJob *J, void *pred_y, uint32_t strideY, void *pred_cb, void *pred_cr,
                         uint32_t strideC)
{
    const int N = (int)J->size, nb = N / 4, maxv = bps == 1 ? 255 : 1023, mid = bps == 1 ? 128 : 512;
    /* availability of the 4N/4 + 1 neighbour groups in scan order: left from the bottom, top-left, top from the left */
    int avail[2 * 16 + 1];
    for (int k = 0; k < 2 * nb; k++) { /* left group k covers rows [2N-4-4k, 2N-4k) */
        const int e = J->mode_left[(2 * N - 4 - 4 * k) >> 2];
        avail[k] = !(e == 0xFE || (!J->bottom_left_ok && k < nb) || e == 0xFF || J->pic_left || (e == 1 && J->constrained_intra));
    }
    avail[2 * nb] = !(J->mode_tl == 0xFF || J->pic_left || J->pic_top || (J->mode_tl == 1 && J->constrained_intra));
    for (int k = 0; k < 2 * nb; k++) {
        const int e = J->mode_top[k];
        avail[2 * nb + 1 + k] = !(e == 0xFE || (!J->top_right_ok && k >= nb) || e == 0xFF || J->pic_top || (J->pic_right && k >= nb) ||
                                  (e == 1 && J->constrained_intra));
    }
    int any = 0;
    for (int k = 0; k < 4 * nb + 1; k++)
        any |= avail[k];
    for (int p = 0; p < 3; p++) {
        if (!(p == 0 ? pred_y : p == 1 ? pred_cb : pred_cr))
            continue; /* plane not asked for (the mode decision predicts luma and the chroma pair separately) */
        const int n = p ? N / 2 : N, g = p ? 2 : 4; /* samples per group in this plane */
        int border[4 * 64 + 1];                      /* scan order: bottom-left ... top-left ... top-right */
        if (!any) {
            for (int i = 0; i < 4 * n + 1; i++)
                border[i] = mid;
        } else {
            /* raw samples in scan order, then the substitution process (8.4.4.2.2) */
            int ok[4 * 64 + 1];
            for (int i = 0; i < 2 * n; i++)
                border[i] = J->left[p][2 * n - 1 - i], ok[i] = avail[i / g];
            border[2 * n] = J->tl[p], ok[2 * n] = avail[2 * nb];
            for (int i = 0; i < 2 * n; i++)
                border[2 * n + 1 + i] = J->top[p][i], ok[2 * n + 1 + i] = avail[2 * nb + 1 + i / g];
            int first = 0;
            while (!ok[first])
                first++;
            for (int i = 0; i < first; i++)
                border[i] = border[first];
            for (int i = first + 1; i < 4 * n + 1; i++)
                if (!ok[i])
                    border[i] = border[i - 1];
        }
        /* r[]: left top-to-bottom, top-left, top (the reference's "reverse" arrays) */
        int r[4 * 64 + 1], rf[4 * 64 + 1];
        for (int i = 0; i < 2 * n; i++)
            r[i] = border[2 * n - 1 - i];
        for (int i = 2 * n; i < 4 * n + 1; i++)
            r[i] = border[i];
        int mode = J->luma_mode;
        const int *use = r;
        if (p == 0) {
            /* filtered copy (Part 3, :636-690): strong smoothing for 32x32 when both sides are flat, else [1 2 1] */
            const int bl = border[0], tlv = border[2 * n], tr = border[4 * n], thr = bps == 1 ? 8 : 32;
            int fb[4 * 64 + 1];
            const int flatL = abs(bl + tlv - 2 * border[n]) < thr, flatT = abs(tlv + tr - 2 * border[3 * n]) < thr;
            if (J->strong_smoothing && n >= 32 && flatL && flatT) {
                const int sh = ilog2((uint32_t)n) + 1;
                fb[0] = bl, fb[2 * n] = tlv, fb[4 * n] = tr;
                for (int i = 1; i < 2 * n; i++) {
                    fb[i] = ((2 * n - i) * bl + i * tlv + n) >> sh;
                    fb[2 * n + i] = ((2 * n - i) * tlv + i * tr + n) >> sh;
                }
            } else {
                fb[0] = border[0], fb[4 * n] = border[4 * n];
                for (int i = 1; i < 4 * n; i++)
                    fb[i] = (border[i - 1] + 2 * border[i] + border[i + 1] + 2) >> 2;
            }
            for (int i = 0; i < 2 * n; i++)
                rf[i] = fb[2 * n - 1 - i];
            for (int i = 2 * n; i < 4 * n + 1; i++)
                rf[i] = fb[i];
            static const int thrTab[5] = {35, 7, 1, 0, 10}; /* intraLumaFilterTable (:60-66) */
            const int dA = abs(mode - 10), dB = abs(mode - 26), dm = dA < dB ? dA : dB;
            if (dm > thrTab[ilog2((uint32_t)n) - 2] && mode != 1 && !J->no_smoothing)
                use = rf;
        } else {
            const int cm = J->chroma_mode; /* EB_INTRA_CHROMA_PLANAR 0, VERTICAL 1, HORIZONTAL 2, DC 3, DM 4 */
            mode = cm == 0 ? 0 : cm == 1 ? 26 : cm == 2 ? 10 : cm == 3 ? 1 : (int)J->luma_mode;
        }
        int dc = 0;
        for (int i = 0; i < n; i++)
            dc += use[i] + use[2 * n + 1 + i];
        dc = (dc + n) >> (ilog2((uint32_t)n) + 1);
        void *dst = p == 0 ? pred_y : p == 1 ? pred_cb : pred_cr;
        const uint32_t stride = p ? strideC : strideY;
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++)
                PUT(dst, (size_t)y * stride + x, pu_predict(mode, n, use, x, y, dc, p == 0, maxv));
    }
}
