/*
 * oracle/svt_oracle_intra.c - TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 * CPU restatement of the 24 intra-prediction leaf kernels (8- and 16-bit) of
 * /root/reference/Source/Lib/C_DEFAULT/EbIntraPrediction_C.c:15-1051.
 *
 * Reference-sample layout (size N): ref[0..2N-1] left column top-to-bottom then
 * bottom-left, ref[2N] top-left, ref[2N+1..4N] top row then top-right.
 * `skip` = 1 predicts even rows only.  mode ids: SVT_ORACLE_INTRA_*.
 */
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
