/* Device code shared by the transform kernels (txfm_kernels.hip) and the fused full-loop kernel
 * (fullloop_kernels.hip): HEVC core matrix, LDS working set, one 1-D forward pass.  Each translation unit gets its own
 * copy of the constant table (statically initialised, no upload needed). */
#ifndef SVT_AMD_TXFM_DEVICE_H
#define SVT_AMD_TXFM_DEVICE_H
#include "svt_amd_internal.h"

#define TX_THREADS 256

/* HEVC core transform matrix (H.265 8.6.4.2); N-point row k = row k*32/N, first N columns */
#define SVT_T32_INIT { \
    {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64}, \
    {90, 90, 88, 85, 82, 78, 73, 67, 61, 54, 46, 38, 31, 22, 13, 4, -4, -13, -22, -31, -38, -46, -54, -61, -67, -73, -78, -82, -85, -88, -90, -90}, \
    {90, 87, 80, 70, 57, 43, 25, 9, -9, -25, -43, -57, -70, -80, -87, -90, -90, -87, -80, -70, -57, -43, -25, -9, 9, 25, 43, 57, 70, 80, 87, 90}, \
    {90, 82, 67, 46, 22, -4, -31, -54, -73, -85, -90, -88, -78, -61, -38, -13, 13, 38, 61, 78, 88, 90, 85, 73, 54, 31, 4, -22, -46, -67, -82, -90}, \
    {89, 75, 50, 18, -18, -50, -75, -89, -89, -75, -50, -18, 18, 50, 75, 89, 89, 75, 50, 18, -18, -50, -75, -89, -89, -75, -50, -18, 18, 50, 75, 89}, \
    {88, 67, 31, -13, -54, -82, -90, -78, -46, -4, 38, 73, 90, 85, 61, 22, -22, -61, -85, -90, -73, -38, 4, 46, 78, 90, 82, 54, 13, -31, -67, -88}, \
    {87, 57, 9, -43, -80, -90, -70, -25, 25, 70, 90, 80, 43, -9, -57, -87, -87, -57, -9, 43, 80, 90, 70, 25, -25, -70, -90, -80, -43, 9, 57, 87}, \
    {85, 46, -13, -67, -90, -73, -22, 38, 82, 88, 54, -4, -61, -90, -78, -31, 31, 78, 90, 61, 4, -54, -88, -82, -38, 22, 73, 90, 67, 13, -46, -85}, \
    {83, 36, -36, -83, -83, -36, 36, 83, 83, 36, -36, -83, -83, -36, 36, 83, 83, 36, -36, -83, -83, -36, 36, 83, 83, 36, -36, -83, -83, -36, 36, 83}, \
    {82, 22, -54, -90, -61, 13, 78, 85, 31, -46, -90, -67, 4, 73, 88, 38, -38, -88, -73, -4, 67, 90, 46, -31, -85, -78, -13, 61, 90, 54, -22, -82}, \
    {80, 9, -70, -87, -25, 57, 90, 43, -43, -90, -57, 25, 87, 70, -9, -80, -80, -9, 70, 87, 25, -57, -90, -43, 43, 90, 57, -25, -87, -70, 9, 80}, \
    {78, -4, -82, -73, 13, 85, 67, -22, -88, -61, 31, 90, 54, -38, -90, -46, 46, 90, 38, -54, -90, -31, 61, 88, 22, -67, -85, -13, 73, 82, 4, -78}, \
    {75, -18, -89, -50, 50, 89, 18, -75, -75, 18, 89, 50, -50, -89, -18, 75, 75, -18, -89, -50, 50, 89, 18, -75, -75, 18, 89, 50, -50, -89, -18, 75}, \
    {73, -31, -90, -22, 78, 67, -38, -90, -13, 82, 61, -46, -88, -4, 85, 54, -54, -85, 4, 88, 46, -61, -82, 13, 90, 38, -67, -78, 22, 90, 31, -73}, \
    {70, -43, -87, 9, 90, 25, -80, -57, 57, 80, -25, -90, -9, 87, 43, -70, -70, 43, 87, -9, -90, -25, 80, 57, -57, -80, 25, 90, 9, -87, -43, 70}, \
    {67, -54, -78, 38, 85, -22, -90, 4, 90, 13, -88, -31, 82, 46, -73, -61, 61, 73, -46, -82, 31, 88, -13, -90, -4, 90, 22, -85, -38, 78, 54, -67}, \
    {64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64}, \
    {61, -73, -46, 82, 31, -88, -13, 90, -4, -90, 22, 85, -38, -78, 54, 67, -67, -54, 78, 38, -85, -22, 90, 4, -90, 13, 88, -31, -82, 46, 73, -61}, \
    {57, -80, -25, 90, -9, -87, 43, 70, -70, -43, 87, 9, -90, 25, 80, -57, -57, 80, 25, -90, 9, 87, -43, -70, 70, 43, -87, -9, 90, -25, -80, 57}, \
    {54, -85, -4, 88, -46, -61, 82, 13, -90, 38, 67, -78, -22, 90, -31, -73, 73, 31, -90, 22, 78, -67, -38, 90, -13, -82, 61, 46, -88, 4, 85, -54}, \
    {50, -89, 18, 75, -75, -18, 89, -50, -50, 89, -18, -75, 75, 18, -89, 50, 50, -89, 18, 75, -75, -18, 89, -50, -50, 89, -18, -75, 75, 18, -89, 50}, \
    {46, -90, 38, 54, -90, 31, 61, -88, 22, 67, -85, 13, 73, -82, 4, 78, -78, -4, 82, -73, -13, 85, -67, -22, 88, -61, -31, 90, -54, -38, 90, -46}, \
    {43, -90, 57, 25, -87, 70, 9, -80, 80, -9, -70, 87, -25, -57, 90, -43, -43, 90, -57, -25, 87, -70, -9, 80, -80, 9, 70, -87, 25, 57, -90, 43}, \
    {38, -88, 73, -4, -67, 90, -46, -31, 85, -78, 13, 61, -90, 54, 22, -82, 82, -22, -54, 90, -61, -13, 78, -85, 31, 46, -90, 67, 4, -73, 88, -38}, \
    {36, -83, 83, -36, -36, 83, -83, 36, 36, -83, 83, -36, -36, 83, -83, 36, 36, -83, 83, -36, -36, 83, -83, 36, 36, -83, 83, -36, -36, 83, -83, 36}, \
    {31, -78, 90, -61, 4, 54, -88, 82, -38, -22, 73, -90, 67, -13, -46, 85, -85, 46, 13, -67, 90, -73, 22, 38, -82, 88, -54, -4, 61, -90, 78, -31}, \
    {25, -70, 90, -80, 43, 9, -57, 87, -87, 57, -9, -43, 80, -90, 70, -25, -25, 70, -90, 80, -43, -9, 57, -87, 87, -57, 9, 43, -80, 90, -70, 25}, \
    {22, -61, 85, -90, 73, -38, -4, 46, -78, 90, -82, 54, -13, -31, 67, -88, 88, -67, 31, 13, -54, 82, -90, 78, -46, 4, 38, -73, 90, -85, 61, -22}, \
    {18, -50, 75, -89, 89, -75, 50, -18, -18, 50, -75, 89, -89, 75, -50, 18, 18, -50, 75, -89, 89, -75, 50, -18, -18, 50, -75, 89, -89, 75, -50, 18}, \
    {13, -38, 61, -78, 88, -90, 85, -73, 54, -31, 4, 22, -46, 67, -82, 90, -90, 82, -67, 46, -22, -4, 31, -54, 73, -85, 90, -88, 78, -61, 38, -13}, \
    {9, -25, 43, -57, 70, -80, 87, -90, 90, -87, 80, -70, 57, -43, 25, -9, -9, 25, -43, 57, -70, 80, -87, 90, -90, 87, -80, 70, -57, 43, -25, 9}, \
    {4, -13, 22, -31, 38, -46, 54, -61, 67, -73, 78, -82, 85, -88, 90, -90, 90, -90, 88, -85, 82, -78, 73, -67, 61, -54, 46, -38, 31, -22, 13, -4}}
static __constant__ int8_t c_T32[32][32] = SVT_T32_INIT;
/* the same matrix as a constant the compiler can fold: fully unrolled code turns its entries into immediates */
static __device__ const int8_t d_T32[32][32] = SVT_T32_INIT;


__device__ __forceinline__ int clip16i(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

/* ------------------------------------------------------------------------- */
/* forward / inverse transforms: one workgroup per group of GB blocks         */
/* ------------------------------------------------------------------------- */

template <int N, int G = ((N >= 16) ? 1 : (N == 8 ? 4 : 16))>
struct TxShared {
    static constexpr int GB = G; /* blocks per workgroup */
    int8_t T[32][32];
    int16_t io[GB][N * N];  /* input, later the transposed first-pass output */
    int32_t E[GB][N][N + 1]; /* running even vector, levels evaluated in place; +1: the dot-product stage walks  */
    int32_t D[GB][N][N + 1]; /* rows with the row index fastest, unpadded rows would all hit one LDS bank.       */
                             /* D: [h..2h) = odd vector of length h */
};

/* one 1-D forward pass over all rows of the GB blocks held in S.io; output transposed into dst */
template <int N, bool TO_GLOBAL, int NT = TX_THREADS, int G = ((N >= 16) ? 1 : (N == 8 ? 4 : 16))>
__device__ void fwd_pass(TxShared<N, G> &S, int shift, int wrap_levels, int16_t *gdst, int nvalid, int t)
{
    constexpr int GB = G;
    constexpr int LOG = N == 32 ? 5 : N == 16 ? 4 : N == 8 ? 3 : 2;
    for (int i = t; i < GB * N * N; i += NT)
        S.E[i / (N * N)][(i / N) % N][i % N] = (&S.io[0][0])[i];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < LOG - 1; m++) {
        const int half = N >> (m + 1);
        for (int i = t; i < GB * N * half; i += NT) {
            const int j = i % half, r = (i / half) % N, b = i / (half * N);
            int s = S.E[b][r][j] + S.E[b][r][2 * half - 1 - j], d = S.E[b][r][j] - S.E[b][r][2 * half - 1 - j];
            if (m < wrap_levels)
                s = (int16_t)s, d = (int16_t)d;
            S.D[b][r][half + j] = d;
            S.E[b][r][j] = s; /* indices >= half are only read in this level: no hazard */
        }
        __syncthreads();
    }
    const int offset = (int16_t)(1 << (shift - 1));
    for (int i = t; i < GB * N * N; i += NT) {
        const int r = i % N, k = (i / N) % N, b = i / (N * N);
        int acc = 0;
        if ((k & (N / 2 - 1)) == 0) { /* k == 0 or N/2: last even pair */
            acc = S.T[k * (32 / N)][0] * S.E[b][r][0] + S.T[k * (32 / N)][1] * S.E[b][r][1];
        } else {
            const int m = __ffs(k) - 1, len = N >> (m + 1);
            const int8_t *c = S.T[k * (32 / N)];
            const int32_t *dv = &S.D[b][r][len];
            for (int j = 0; j < len; j++)
                acc += c[j] * dv[j];
        }
        const int16_t v = (int16_t)((acc + offset) >> shift);
        if (TO_GLOBAL) {
            if (b < nvalid)
                gdst[(size_t)b * N * N + k * N + r] = v;
        } else {
            S.io[b][k * N + r] = v;
        }
    }
    __syncthreads();
}


/* ------------------------------------------------------------------------- */
/* register-resident forward transform: one lane = one row                   */
/* ------------------------------------------------------------------------- */
/* One 1-D N-point forward pass of the partial-butterfly transforms (C_DEFAULT/EbTransforms_C.c:1602-1861) on a row held
 * in registers: e[] = the row (destroyed), emit(k, v) receives output k.  Same arithmetic as fwd_pass above (even / odd
 * split per level, the low-precision "Estimate" variants wrap the first `wrap_levels` levels to 16 bits); loops are
 * fully unrolled so every matrix entry is an immediate and no coefficient is ever read from memory. */
template <int N, typename Emit>
__device__ __forceinline__ void fwd_1d_regs(int (&e)[N], int shift, int wrap_levels, Emit emit)
{
    constexpr int LOG = N == 32 ? 5 : N == 16 ? 4 : N == 8 ? 3 : 2;
    int d[N]; /* [half, 2*half) = odd vector of the level whose half-length is `half` */
#pragma unroll
    for (int m = 0; m < LOG - 1; m++) {
        const int half = N >> (m + 1);
#pragma unroll
        for (int j = 0; j < half; j++) {
            int s = e[j] + e[2 * half - 1 - j], dd = e[j] - e[2 * half - 1 - j];
            if (m < wrap_levels)
                s = (int16_t)s, dd = (int16_t)dd;
            d[half + j] = dd;
            e[j] = s;
        }
    }
    const int offset = (int16_t)(1 << (shift - 1));
#pragma unroll
    for (int k = 0; k < N; k++) {
        int acc = 0;
        if ((k & (N / 2 - 1)) == 0) {
            acc = d_T32[k * (32 / N)][0] * e[0] + d_T32[k * (32 / N)][1] * e[1];
        } else {
            const int m = __builtin_ctz(k), len = N >> (m + 1);
#pragma unroll
            for (int j = 0; j < len; j++)
                acc += d_T32[k * (32 / N)][j] * d[len + j];
        }
        emit(k, (int16_t)((acc + offset) >> shift));
    }
}

/* One 1-D N-point INVERSE pass (InvTransform32x32 .. 4x4, C_DEFAULT/EbTransforms_C.c:1910-2094) on coefficients held in
 * registers: exact 32-bit even / odd recombination (equal to the plain matrix product), v[j] = sum_k T[k][j] * c[k]. */
template <int N, int L, int S>
struct InvBF {
    static __device__ __forceinline__ void run(const int (&c)[N], int (&v)[L])
    {
        int e[L / 2];
        InvBF<N, L / 2, 2 * S>::run(c, e);
#pragma unroll
        for (int k = 0; k < L / 2; k++) {
            int o = 0;
#pragma unroll
            for (int i = 1; i < L; i += 2)
                o += d_T32[(i * S) * (32 / N)][k] * c[i * S];
            v[k] = e[k] + o;
            v[L - 1 - k] = e[k] - o;
        }
    }
};
template <int N, int S>
struct InvBF<N, 2, S> {
    static __device__ __forceinline__ void run(const int (&c)[N], int (&v)[2])
    {
        v[0] = d_T32[0][0] * c[0] + d_T32[S * (32 / N)][0] * c[S];
        v[1] = d_T32[0][1] * c[0] + d_T32[S * (32 / N)][1] * c[S];
    }
};
template <int N, typename Emit>
__device__ __forceinline__ void inv_1d_regs(const int (&c)[N], int shift, Emit emit)
{
    int v[N];
    InvBF<N, N, 1>::run(c, v);
    const int offset = (int16_t)(1 << (shift - 1));
#pragma unroll
    for (int j = 0; j < N; j++)
        emit(j, (int16_t)clip16i((v[j] + offset) >> shift));
}

/* A row of N samples (8- or 16-bit) <-> registers, with 16-byte (or whole-row) vector accesses when the row is aligned */
template <int N, typename T>
__device__ __forceinline__ void load_row(const T *p, int (&v)[N])
{
    constexpr int BYTES = N * (int)sizeof(T), VB = BYTES >= 16 ? 16 : BYTES; /* vector width in bytes: 16, 8 or 4 */
    if (((uintptr_t)p & (VB - 1)) == 0) {
#pragma unroll
        for (int o = 0; o < BYTES; o += VB) {
            uint32_t w[4];
            if (VB == 16) {
                const uint4 q = *(const uint4 *)((const uint8_t *)p + o);
                w[0] = q.x, w[1] = q.y, w[2] = q.z, w[3] = q.w;
            } else if (VB == 8) {
                const uint2 q = *(const uint2 *)((const uint8_t *)p + o);
                w[0] = q.x, w[1] = q.y;
            } else {
                w[0] = *(const uint32_t *)((const uint8_t *)p + o);
            }
#pragma unroll
            for (int k = 0; k < VB / (int)sizeof(T); k++) {
                const int bit = k * 8 * (int)sizeof(T);
                v[o / (int)sizeof(T) + k] = (int)((w[bit >> 5] >> (bit & 31)) & (sizeof(T) == 1 ? 0xffu : 0xffffu));
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; j++)
            v[j] = (int)p[j];
    }
}
template <int N, typename T>
__device__ __forceinline__ void store_row(T *p, const int (&v)[N])
{
    constexpr int BYTES = N * (int)sizeof(T), VB = BYTES >= 16 ? 16 : BYTES;
    if (((uintptr_t)p & (VB - 1)) == 0) {
#pragma unroll
        for (int o = 0; o < BYTES; o += VB) {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < VB / (int)sizeof(T); k++) {
                const int bit = k * 8 * (int)sizeof(T);
                w[bit >> 5] |= ((uint32_t)v[o / (int)sizeof(T) + k] & (sizeof(T) == 1 ? 0xffu : 0xffffu)) << (bit & 31);
            }
            if (VB == 16)
                *(uint4 *)((uint8_t *)p + o) = make_uint4(w[0], w[1], w[2], w[3]);
            else if (VB == 8)
                *(uint2 *)((uint8_t *)p + o) = make_uint2(w[0], w[1]);
            else
                *(uint32_t *)((uint8_t *)p + o) = w[0];
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; j++)
            p[j] = (T)v[j];
    }
}

/* LDS tile of the register transform: the intermediate of one N x N unit, rows padded by two samples so that a lane
 * per row and a lane per column both walk distinct banks; units 16 dwords apart from a bank-aligned pitch */
template <int N>
struct TxRegTile {
    static constexpr int PITCH = N + 2;                       /* int16 per row */
    static constexpr int UNIT = N * PITCH + 32;               /* int16 per unit */
};

/* Two passes of one unit by the N lanes that own it (lane `r` of the unit: row r in pass 1, column r of the result in
 * pass 2).  x[] = row r of the residual on entry; on return x[j] = coefficient (j, r) of the transform, i.e. every lane
 * holds one COLUMN of the coefficient block.  tile: this unit's TxRegTile<N>::UNIT int16 of LDS.  All lanes of the
 * unit are in one wave; the LDS exchange needs no workgroup barrier. */
template <int N>
__device__ __forceinline__ void fwd_2d_regs(int (&x)[N], int16_t *tile, int r, int shift1, int shift2, int wrap_levels)
{
    constexpr int P = TxRegTile<N>::PITCH;
    fwd_1d_regs<N>(x, shift1, wrap_levels, [&](int k, int16_t v) { tile[k * P + r] = v; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < N; j += 2) {
        const uint32_t w = *(const uint32_t *)&tile[r * P + j];
        x[j] = (int16_t)(w & 0xffffu), x[j + 1] = (int16_t)(w >> 16);
    }
    int y[N];
    fwd_1d_regs<N>(x, shift2, wrap_levels, [&](int k, int16_t v) { y[k] = v; });
#pragma unroll
    for (int j = 0; j < N; j++)
        x[j] = y[j];
}

#endif
