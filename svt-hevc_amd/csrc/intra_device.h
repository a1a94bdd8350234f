/* H.265 8.4.4.2 sample prediction shared by the prediction-unit kernel (intra_kernels.hip) and the device-resident encode pass
 * (encdec_kernels.hip): one predicted sample from the reference array of its unit. */
#ifndef SVT_AMD_INTRA_DEVICE_H
#define SVT_AMD_INTRA_DEVICE_H
#include "svt_amd_internal.h"

/* intraPredAngle {0, 2, 5, 9, 13, 17, 21, 26, 32} and invAngle {-, 4096, 1638, 910, 630, 482, 390, 315, 256} by |distance| from the
 * horizontal / vertical mode, as packed immediates: a table in memory is a dependent HBM / L2 round trip per unit on the encode
 * pass's critical path */
__device__ __forceinline__ int pu_ang(int d)
{
    const unsigned long long k = 0ull | (2ull << 6) | (5ull << 12) | (9ull << 18) | (13ull << 24) | (17ull << 30) | (21ull << 36) | (26ull << 42) | (32ull << 48);
    return (int)((k >> (6 * d)) & 63);
}
__device__ __forceinline__ int pu_inv(int d) /* d in 1..8 */
{
    const unsigned long long lo = 4096ull | (1638ull << 16) | (910ull << 32) | (630ull << 48), hi = 482ull | (390ull << 16) | (315ull << 32) | (256ull << 48);
    return (int)(((d <= 4 ? lo : hi) >> (16 * ((d - 1) & 3))) & 0xffff);
}

/* r: left[0..2N-1] top to bottom, r[2N] top-left, r[2N+1+j] top[j] */
__device__ __forceinline__ int pu_predict(int mode, int N, int lg, const int16_t *r, int x, int y, int dc, bool lumaEdge, int maxv)
{
    const int16_t *left = r, *top = r + 2 * N + 1;
    const int tl = r[2 * N];
    if (mode == 0)
        return ((N - 1 - x) * left[y] + (x + 1) * top[N] + (N - 1 - y) * top[x] + (y + 1) * left[N] + N) >> (lg + 1);
    if (mode == 1) {
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
        return (lumaEdge && N < 32 && x == 0) ? min(maxv, max(0, top[0] + ((left[y] - tl) >> 1))) : (int)top[x];
    if (mode == 10)
        return (lumaEdge && N < 32 && y == 0) ? min(maxv, max(0, left[0] + ((top[x] - tl) >> 1))) : (int)left[y];
    const bool vert = mode >= 18;
    const int d = vert ? mode - 26 : 10 - mode;
    const int a = d < 0 ? -pu_ang(-d) : pu_ang(d);
    const int inv = d < 0 ? pu_inv(-d) : 0;
    const int u = vert ? x : y, v = vert ? y : x;
    const int16_t *mainr = vert ? top : left, *side = vert ? left : top;
    const int pos = (v + 1) * a, i = pos >> 5, f = pos & 31;
    int s[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int idx = u + i + 1 + k;
        s[k] = idx > 0 ? mainr[idx - 1] : idx == 0 ? tl : side[((-idx * inv + 128) >> 8) - 1];
    }
    return ((32 - f) * s[0] + f * s[1] + 16) >> 5;
}

/* FOUR samples at once - (x .. x + 3, y) for the planar, DC and vertical-class modes (0, 1, 18..34), (x, y .. y + 3) for the horizontal-class modes (2..17): the samples of a
 * run share the row's (column's) projection and weights, and a wave that predicts a sample per lane and step spends its time on exactly that set-up (the mode decision's
 * loops are bound by instructions issued per wave).  Same values as pu_predict sample for sample; x (y) a multiple of 4, N >= 4. */
__device__ __forceinline__ bool pu_horizontal_class(int mode) { return mode >= 2 && mode < 18; }
__device__ __forceinline__ void pu_predict4(int mode, int N, int lg, const int16_t *r, int x, int y, int dc, bool lumaEdge, int maxv, int (&o)[4])
{
    const int16_t *left = r, *top = r + 2 * N + 1;
    const int tl = r[2 * N];
    if (mode == 0) {
        const int l = left[y], tn = top[N], ln = left[N];
        const int base = (N - 1 - y) , c0 = (y + 1) * ln + N;
#pragma unroll
        for (int k = 0; k < 4; k++)
            o[k] = ((N - 1 - (x + k)) * l + (x + k + 1) * tn + base * top[x + k] + c0) >> (lg + 1);
        return;
    }
    if (mode == 1) {
        const bool edge = lumaEdge && N < 32;
#pragma unroll
        for (int k = 0; k < 4; k++)
            o[k] = dc;
        if (edge) {
            if (y == 0) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    o[k] = (top[x + k] + 3 * dc + 2) >> 2;
                if (x == 0)
                    o[0] = (left[0] + top[0] + 2 * dc + 2) >> 2;
            } else if (x == 0) {
                o[0] = (left[y] + 3 * dc + 2) >> 2;
            }
        }
        return;
    }
    if (mode == 26) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            o[k] = top[x + k];
        if (lumaEdge && N < 32 && x == 0)
            o[0] = min(maxv, max(0, top[0] + ((left[y] - tl) >> 1)));
        return;
    }
    if (mode == 10) { /* (horizontal class: a run down the column) */
#pragma unroll
        for (int k = 0; k < 4; k++)
            o[k] = left[y + k];
        if (lumaEdge && N < 32 && y == 0)
            o[0] = min(maxv, max(0, left[0] + ((top[x] - tl) >> 1)));
        return;
    }
    const bool vert = mode >= 18;
    const int d = vert ? mode - 26 : 10 - mode;
    const int a = d < 0 ? -pu_ang(-d) : pu_ang(d);
    const int inv = d < 0 ? pu_inv(-d) : 0;
    const int u = vert ? x : y, v = vert ? y : x; /* the run goes along u */
    const int16_t *mainr = vert ? top : left, *side = vert ? left : top;
    const int pos = (v + 1) * a, i = pos >> 5, f = pos & 31;
    int s[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int idx = u + i + 1 + k;
        s[k] = idx > 0 ? mainr[idx - 1] : idx == 0 ? tl : side[((-idx * inv + 128) >> 8) - 1];
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
        o[k] = ((32 - f) * s[k] + f * s[k + 1] + 16) >> 5;
}

#endif
