/*
 * In-loop filter and bit-depth packing kernels (EncDec rows "BS / DLF", "SAO", "pack/unpack").
 *
 *   deblocking edge cores  Luma4SampleEdgeDLFCore(16bit), Chroma2SampleEdgeDLFCore(16bit)
 *                          (C_DEFAULT/EbDeblockingFilter_C.c:39-577; tables Codec/EbDeblockingFilter.h:245-272)
 *                          batched: one thread per 4-sample (luma) / 2-sample (chroma) edge of an edge list;
 *                          the edges of one launch must not overlap (all vertical edges of a picture, then
 *                          all horizontal ones - exactly the order the reference's LCU drivers use).
 *   SAO statistics         GatherSaoStatisticsLcu* (C_DEFAULT/EbSampleAdaptiveOffset_C.c:23-393; tables
 *                          Codec/EbSampleAdaptiveOffset.h:171-206): one workgroup per LCU, LDS histograms.
 *   SAO apply              SAOApplyBO / SAOApplyEO_0/90/135/45 (+16bit) (:394-861; tables :209-360): one thread
 *                          per sample, classified against the ORIGINAL neighbours (out of place), which is what
 *                          the reference's in-place sign-carrying loops compute.
 *   pack / unpack          EB_ENC_msbPack2D, CompressedPackmsb, CPack_C, EB_ENC_msbUnPack2D, UnPack8BitData,
 *                          UnpackAvg (C_DEFAULT/EbPackUnPack_C.c:12-251; tables Codec/EbPackUnPack.h:27-175):
 *                          pure streaming, HBM-bound.
 */
#include "leaf_util.h"
#include <cstring>
#include <type_traits>
#include <cstdlib>

__device__ __forceinline__ int f_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int f_sgn(int a, int b) { return (a - b) < 0 ? -1 : ((a - b) > 0 ? 1 : 0); }

/* ---------------- deblocking ---------------- */
struct DlfLumaEdge { int32_t offset; int16_t tc, beta; uint8_t vertical, pad[3]; };     /* = SvtAmdDlfLumaEdge */
struct DlfChromaEdge { int32_t offset; uint8_t cb_tc, cr_tc, vertical, pad; };          /* = SvtAmdDlfChromaEdge */

template <typename T>
__device__ void dlf_luma_core(T *edge, int stride, int vertical, int tc, int beta)
{
    const int maxv = sizeof(T) == 1 ? 255 : 1023;
    const int fs = vertical ? 1 : stride, ns = vertical ? stride : 1;
#define S(k, line) ((int)edge[(k) * fs + (line) * ns])
    const int dp0 = abs(S(-3, 0) - 2 * S(-2, 0) + S(-1, 0)), dp3 = abs(S(-3, 3) - 2 * S(-2, 3) + S(-1, 3));
    const int dq0 = abs(S(2, 0) - 2 * S(1, 0) + S(0, 0)), dq3 = abs(S(2, 3) - 2 * S(1, 3) + S(0, 3));
    const int dp = dp0 + dp3, dq = dq0 + dq3, d0 = dp0 + dq0, d3 = dp3 + dq3, d = d0 + d3;
    if (d >= beta)
        return;
    bool strong = true;
#pragma unroll
    for (int line = 0; line < 4; line += 3) {
        const int dl = line ? d3 : d0;
        strong = strong && ((dl << 1) < (beta >> 2)) &&
                 (beta >> 3) > (abs(S(-4, line) - S(-1, line)) + abs(S(3, line) - S(0, line))) &&
                 ((5 * tc + 1) >> 1) > abs(S(-1, line) - S(0, line));
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int q0 = S(0, c), q1 = S(1, c), q2 = S(2, c), q3 = S(3, c);
        const int p0 = S(-1, c), p1 = S(-2, c), p2 = S(-3, c), p3 = S(-4, c);
#define W(k, v) edge[(k) * fs + c * ns] = (T)(v)
        if (strong) {
            W(0, f_clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3));
            W(-1, f_clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3));
            W(1, f_clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2));
            W(-2, f_clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2));
            W(2, f_clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3));
            W(-3, f_clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3));
        } else {
            int delta = ((q0 - p0) * 9 - (q1 - p1) * 3 + 8) >> 4;
            if (abs(delta) < tc * 10) {
                delta = f_clip3(-tc, tc, delta);
                W(0, f_clip3(0, maxv, q0 - delta));
                W(-1, f_clip3(0, maxv, p0 + delta));
                const int side = (beta + (beta >> 1)) >> 3, tc2 = tc >> 1;
                if (side > dp)
                    W(-2, f_clip3(0, maxv, p1 + f_clip3(-tc2, tc2, ((((p0 + p2 + 1) >> 1) - p1 + delta) >> 1))));
                if (side > dq)
                    W(1, f_clip3(0, maxv, q1 + f_clip3(-tc2, tc2, ((((q0 + q2 + 1) >> 1) - q1 - delta) >> 1))));
            }
        }
#undef W
    }
#undef S
}

template <typename T>
__global__ void k_dlf_luma(T *plane, int stride, const DlfLumaEdge *edges, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dlf_luma_core<T>(plane + edges[i].offset, stride, edges[i].vertical, edges[i].tc, edges[i].beta);
}

template <typename T>
__global__ void k_dlf_chroma(T *cb, T *cr, int stride, const DlfChromaEdge *edges, uint32_t n)
{
    const int maxv = sizeof(T) == 1 ? 255 : 1023;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int fs = edges[i].vertical ? 1 : stride, ns = edges[i].vertical ? stride : 1;
#pragma unroll
        for (int plane = 0; plane < 2; plane++) {
            T *e = (plane ? cr : cb) + edges[i].offset;
            const int tc = plane ? edges[i].cr_tc : edges[i].cb_tc;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int q0 = e[c * ns], q1 = e[c * ns + fs], p0 = e[c * ns - fs], p1 = e[c * ns - 2 * fs];
                const int delta = (int16_t)f_clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
                e[c * ns - fs] = (T)f_clip3(0, maxv, p0 + delta);
                e[c * ns] = (T)f_clip3(0, maxv, q0 - delta);
            }
        }
    }
}

/* ---------------- deblocking, whole picture ---------------- */
/* What LCUInternalAreaDLFCore / LCUBoundaryDLFCore / LCUPictureEdgeDLFCore (+16bit; EbDeblockingFilter.c:2222-4330) leave
 * behind once they have run over every LCU: all vertical edges of the 8x8 grid (launch dir = 0), then all horizontal
 * ones (dir = 1).  One thread per 4-sample luma segment or 2-sample chroma segment; strength from the per-LCU arrays,
 * tc / beta from the mean qp of the two sides.  Within one direction the segments touch disjoint samples, so the
 * filter runs in place; neighbouring threads own neighbouring 8-byte (vertical) / 4-byte (horizontal) groups of a row,
 * so every row access of a wave is one contiguous run. */
struct DlfPic {
    void *y, *cb, *cr;
    const uint8_t *bs_v, *bs_h, *qp;
    int strideY, strideC, width, height, qpStride, lcuCols;
    int tcOffset, betaOffset, cbQpOffset, crQpOffset;
};
__constant__ uint8_t c_dlf_tc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                     2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
__constant__ uint8_t c_dlf_beta[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                       16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
__constant__ uint8_t c_chroma_qp_map[58] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28,
                                            29, 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51};

__device__ __forceinline__ int dlf_chroma_tc(int qpMean, int qpOffset, int tcOffset)
{
    const int q = qpMean + qpOffset;
    const uint8_t qc = (uint8_t)(q < 0 ? q : q > 57 ? q - 6 : c_chroma_qp_map[q]); /* convertToChromaQp into an EB_U8 (:21-25) */
    return c_dlf_tc[f_clip3(0, 53, (int)qc + 2 + tcOffset)];
}

template <typename T>
__global__ __launch_bounds__(256) void k_dlf_picture(const DlfPic P, int dir, uint32_t nLumaX, uint32_t nLuma, uint32_t nChromaX,
                                                     uint32_t nChroma)
{
    const int scale = sizeof(T) == 1 ? 0 : 2, maxv = sizeof(T) == 1 ? 255 : 1023;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nLuma + nChroma; i += gridDim.x * blockDim.x) {
        if (i < nLuma) {
            const uint32_t sy = i / nLumaX, sx = i - sy * nLumaX;
            const int px = dir ? (int)sx * 4 : (int)(sx + 1) * 8, py = dir ? (int)(sy + 1) * 8 : (int)sy * 4;
            const uint8_t *bs = dir ? P.bs_h : P.bs_v;
            const int b = bs[((py >> 6) * P.lcuCols + (px >> 6)) * 256 + ((px & 63) >> 2) + (((py & 63) >> 2) << 4)];
            if (!b)
                continue;
            const int qq = P.qp[(px >> 3) + (py >> 3) * P.qpStride];
            const int qpp = dir ? P.qp[(px >> 3) + ((py - 1) >> 3) * P.qpStride] : P.qp[((px - 1) >> 3) + (py >> 3) * P.qpStride];
            const int Q = (qq + qpp + 1) >> 1;
            const int tc = c_dlf_tc[f_clip3(0, 53, Q + ((b > 1) << 1) + P.tcOffset)] << scale;
            const int beta = c_dlf_beta[f_clip3(0, 51, Q + P.betaOffset)] << scale;
            dlf_luma_core<T>((T *)P.y + (size_t)py * P.strideY + px, P.strideY, !dir, tc, beta);
        } else {
            const uint32_t j = i - nLuma, sy = j / nChromaX, sx = j - sy * nChromaX;
            const int cx = dir ? (int)sx * 2 : (int)(sx + 1) * 8, cy = dir ? (int)(sy + 1) * 8 : (int)sy * 2;
            const uint8_t *bs = dir ? P.bs_h : P.bs_v;
            const int b = bs[((cy >> 5) * P.lcuCols + (cx >> 5)) * 256 + ((cx & 31) >> 1) + (((cy & 31) >> 1) << 4)];
            if (b <= 1)
                continue;
            const int qq = P.qp[((2 * cx) >> 3) + ((2 * cy) >> 3) * P.qpStride];
            const int qpp = dir ? P.qp[((2 * cx) >> 3) + ((2 * (cy - 1)) >> 3) * P.qpStride]
                                : P.qp[((2 * (cx - 1)) >> 3) + ((2 * cy) >> 3) * P.qpStride];
            const int Q = (qq + qpp + 1) >> 1;
            const int fs = dir ? P.strideC : 1, ns = dir ? 1 : P.strideC;
#pragma unroll
            for (int plane = 0; plane < 2; plane++) {
                T *e = (T *)(plane ? P.cr : P.cb) + (size_t)cy * P.strideC + cx;
                const int tc = (uint8_t)(dlf_chroma_tc(Q, plane ? P.crQpOffset : P.cbQpOffset, P.tcOffset) << scale);
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const int q0 = e[c * ns], q1 = e[c * ns + fs], p0 = e[c * ns - fs], p1 = e[c * ns - 2 * fs];
                    const int delta = (int16_t)f_clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
                    e[c * ns - fs] = (T)f_clip3(0, maxv, p0 + delta);
                    e[c * ns] = (T)f_clip3(0, maxv, q0 - delta);
                }
            }
        }
    }
}

/* ---------------- boundary strengths, whole picture ---------------- */
/* SetBSArrayBasedOnPUBoundary / SetBSArrayBasedOnTUBoundary with CalculateBSForPUBoundary (Codec/EbDeblockingFilter.c:109-530),
 * which the encode pass runs per coding unit against its neighbour arrays, as one data-parallel pass over picture-level
 * maps: one thread per (direction, 8x8 block) decides the two 4-sample segments of the block's left / top side. */
struct CuMapEntry { uint8_t mode, dir, size_log2, pad; int16_t mv[2][2]; }; /* = SvtAmdCuMapEntry */

__device__ __forceinline__ bool bs_mv_far(const int16_t *a, const int16_t *b) { return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= 4; }

__global__ __launch_bounds__(256) void k_bs_picture(const CuMapEntry *__restrict__ map, const uint8_t *__restrict__ cbf, int width, int height,
                                                   int sliceType, unsigned long long poc0, unsigned long long poc1,
                                                   const uint8_t *__restrict__ lcuEdge, uint8_t *__restrict__ bs_v, uint8_t *__restrict__ bs_h)
{
    const int bw = width >> 3, bh = height >> 3, cw = width >> 2, lcuCols = (width + 63) >> 6;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * bw * bh; i += gridDim.x * blockDim.x) {
        const int dir = i >= bw * bh, j = dir ? i - bw * bh : i, by = j / bw, bx = j - by * bw;
        const CuMapEntry cur = map[by * bw + bx];
        const int x = bx << 3, y = by << 3, size = 1 << cur.size_log2, pos = dir ? y : x;
        const int lcu = (y >> 6) * lcuCols + (x >> 6);
        uint8_t *out = (dir ? bs_h : bs_v) + lcu * 256;
        const bool cuEdge = (pos & (size - 1)) == 0, tuEdge = !cuEdge && size == 64 && (pos & 31) == 0;
        int b[2] = {0, 0};
        if ((cuEdge || tuEdge) && pos != 0 && !(cuEdge && (pos & 63) == 0 && (lcuEdge[lcu] & (dir ? 2 : 1)))) {
            const CuMapEntry nb = dir ? map[(by - 1) * bw + bx] : map[by * bw + bx - 1];
            int c1 = 1;
            if (!tuEdge && cur.mode != 2 && nb.mode != 2) {
                if (sliceType == 1) {
                    c1 = bs_mv_far(cur.mv[0], nb.mv[0]);
                } else {
                    switch (cur.dir + nb.dir * 3) {
                    case 0: c1 = bs_mv_far(cur.mv[0], nb.mv[0]); break;
                    case 1: c1 = poc1 != poc0 || bs_mv_far(cur.mv[1], nb.mv[0]); break;
                    case 3: c1 = poc0 != poc1 || bs_mv_far(cur.mv[0], nb.mv[1]); break;
                    case 4: c1 = bs_mv_far(cur.mv[1], nb.mv[1]); break;
                    case 8:
                        c1 = poc0 == poc1 ? (bs_mv_far(cur.mv[0], nb.mv[0]) || bs_mv_far(cur.mv[1], nb.mv[1])) &&
                                                (bs_mv_far(cur.mv[0], nb.mv[1]) || bs_mv_far(cur.mv[1], nb.mv[0]))
                                          : (bs_mv_far(cur.mv[0], nb.mv[0]) || bs_mv_far(cur.mv[1], nb.mv[1]));
                        break;
                    default: c1 = 1; break;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int sx = dir ? x + 4 * k : x, sy = dir ? y : y + 4 * k, nx = dir ? sx : sx - 4, ny = dir ? sy - 4 : sy;
                const int cbfAny = cbf[(sy >> 2) * cw + (sx >> 2)] || cbf[(ny >> 2) * cw + (nx >> 2)];
                if (tuEdge)
                    b[k] = cur.mode == 2 ? 2 : cbfAny;
                else
                    b[k] = (cur.mode == 2 || nb.mode == 2) ? 2 : (c1 | cbfAny);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int sx = dir ? x + 4 * k : x, sy = dir ? y : y + 4 * k;
            out[((sx & 63) >> 2) + (((sy & 63) >> 2) << 4)] = (uint8_t)b[k];
        }
    }
}

/* ---------------- SAO statistics ---------------- */
struct SaoStats { int32_t boDiff[32]; uint16_t boCount[32]; int32_t eoDiff[4][5]; uint16_t eoCount[4][5]; }; /* = SvtAmdSaoStats */

/* one workgroup per LCU of a picture-wide grid (lcu_size x lcu_size, clipped at the right/bottom) */
template <typename T>
__global__ __launch_bounds__(256) void k_sao_gather(const T *__restrict__ input, int inStride,
                                                    const T *__restrict__ recon, int reconStride, int width, int height,
                                                    int lcu_size, int lcus_w, int only_eo, SaoStats *__restrict__ out)
{
    __shared__ int bo_d[32], eo_d[4][5];
    __shared__ unsigned bo_c[32], eo_c[4][5];
    const int t = threadIdx.x, lcu = blockIdx.x;
    const int x0 = (lcu % lcus_w) * lcu_size, y0 = (lcu / lcus_w) * lcu_size;
    const int lw = min(lcu_size, width - x0), lh = min(lcu_size, height - y0);
    if (t < 32)
        bo_d[t] = 0, bo_c[t] = 0;
    if (t < 20)
        (&eo_d[0][0])[t] = 0, (&eo_c[0][0])[t] = 0;
    __syncthreads();
    const int boShift = sizeof(T) == 1 ? 3 : 5;
    const int iw = lw - 2, ih = lh - 2;
    /* edge-offset histograms live in registers (select-accumulate, compile-time indices): 20 LDS atomics per sample
     * on five hot addresses serialise; band-offset bins are spread by the sample value and stay LDS atomics */
    int ed[4][5];
    unsigned ec[4][5];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int c = 0; c < 5; c++)
            ed[k][c] = 0, ec[k][c] = 0;
    /* LCU-sized areas are staged in LDS with coalesced row loads (the statistics read every sample nine times) */
    __shared__ T tile_r[64 * 64], tile_i[64 * 64];
    const bool staged = lw <= 64 && lh <= 64;
    const T *rp = recon + (ptrdiff_t)y0 * reconStride + x0, *ip = input + (ptrdiff_t)y0 * inStride + x0;
    int rs = reconStride, is = inStride;
    if (staged) {
        /* dword loads where rows are dword-aligned (LCU origins are; picture strides almost always): a quarter (8-bit) or half
         * (16-bit) of the load instructions of the per-sample form */
        constexpr int PER = 4 / (int)sizeof(T);
        const bool vec = (((uintptr_t)rp | (uintptr_t)ip) & 3) == 0 && ((reconStride * (int)sizeof(T)) & 3) == 0 &&
                         ((inStride * (int)sizeof(T)) & 3) == 0 && (lw % PER) == 0;
        if (vec) {
            const int wq = lw / PER;
            for (int i = t; i < wq * lh; i += 256) {
                const int yy = i / wq, xq = i - yy * wq;
                *(uint32_t *)&tile_r[yy * 64 + xq * PER] = *(const uint32_t *)&rp[(ptrdiff_t)yy * reconStride + xq * PER];
                *(uint32_t *)&tile_i[yy * 64 + xq * PER] = *(const uint32_t *)&ip[(ptrdiff_t)yy * inStride + xq * PER];
            }
        } else {
            for (int i = t; i < lw * lh; i += 256) {
                const int yy = i / lw, xx = i - yy * lw;
                tile_r[yy * 64 + xx] = rp[(ptrdiff_t)yy * reconStride + xx];
                tile_i[yy * 64 + xx] = ip[(ptrdiff_t)yy * inStride + xx];
            }
        }
        __syncthreads();
        rp = tile_r, ip = tile_i, rs = 64, is = 64;
    }
    const int total = iw * ih;
    const bool small = iw >= 1 && iw <= 64 && total <= 64 * 64;
    const uint32_t rcw = small ? (1u << 20) / (uint32_t)iw + 1u : 0u; /* i / iw == (i * rcw) >> 20 for i < 4352, iw <= 64 (checked exhaustively) */
    for (int i0 = 0; i0 < total; i0 += 256) { /* uniform trip count: whole waves take part in the shuffles below */
        const int i = i0 + t;
        const bool live = i < total;
        const int q = small ? (int)(((uint32_t)i * rcw) >> 20) : i / iw;
        const int yy = live ? q + 1 : 1, xx = live ? i - q * iw + 1 : 1;
        const T *r = rp + yy * rs + xx;
        const int c = r[0];
        int diff = (int)ip[yy * is + xx] - c;
        if (sizeof(T) == 1)
            diff = f_clip3(-128, 127, diff);
        if (!live)
            diff = 0;
        if (!only_eo) {
            /* smooth content puts a whole wave into one band: then one atomic per wave instead of 64 colliding ones */
            const int bin = c >> boShift, first = __builtin_amdgcn_readfirstlane(bin);
            if (__all(!live || bin == first)) {
                int dsum = diff;
                unsigned csum = live ? 1u : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
                    dsum += __shfl_xor(dsum, o), csum += __shfl_xor(csum, o);
                if ((t & 63) == 0 && csum) {
                    atomicAdd(&bo_d[first], dsum);
                    atomicAdd(&bo_c[first], csum);
                }
            } else if (live) {
                atomicAdd(&bo_d[bin], diff);
                atomicAdd(&bo_c[bin], 1u);
            }
        }
        if (!live)
            continue;
        const int nb[4][2] = {{-1, 1}, {-rs, rs}, {-rs - 1, rs + 1}, {-rs + 1, rs - 1}};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (only_eo && k == 0)
                continue;
            const int idx = f_sgn(c, r[nb[k][0]]) + f_sgn(c, r[nb[k][1]]) + 2;
#pragma unroll
            for (int cc = 0; cc < 5; cc++) {
                ed[k][cc] += idx == cc ? diff : 0;
                ec[k][cc] += idx == cc ? 1u : 0u;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int cc = 0; cc < 5; cc++) {
            int dsum = ed[k][cc];
            unsigned csum = ec[k][cc];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                dsum += __shfl_xor(dsum, o), csum += __shfl_xor(csum, o);
            if ((t & 63) == 0) {
                atomicAdd(&eo_d[k][cc], dsum);
                atomicAdd(&eo_c[k][cc], csum);
            }
        }
    __syncthreads();
    SaoStats *o = &out[lcu];
    if (t < 32 && !only_eo)
        o->boDiff[t] = bo_d[t], o->boCount[t] = (uint16_t)bo_c[t];
    if (t < 4) { /* category compaction: slot 2 <- 3, 3 <- 4 (EbSampleAdaptiveOffset_C.c:113-118) */
        o->eoDiff[t][0] = eo_d[t][0], o->eoDiff[t][1] = eo_d[t][1], o->eoDiff[t][2] = eo_d[t][3];
        o->eoDiff[t][3] = eo_d[t][4], o->eoDiff[t][4] = eo_d[t][4];
        o->eoCount[t][0] = (uint16_t)eo_c[t][0], o->eoCount[t][1] = (uint16_t)eo_c[t][1], o->eoCount[t][2] = (uint16_t)eo_c[t][3];
        o->eoCount[t][3] = (uint16_t)eo_c[t][4], o->eoCount[t][4] = (uint16_t)eo_c[t][4];
    }
}

/* ---------------- SAO apply (leaf level, out of place) ---------------- */
/* src: original samples incl. one row below and one column right; dst: output area. */
template <typename T>
__global__ void k_sao_apply(int kind /* 0..3 EO type, 4 BO */, const T *__restrict__ src, T *__restrict__ dst, int stride,
                            const T *__restrict__ left, const T *__restrict__ upper /* index -1..W */, int band,
                            const int8_t *__restrict__ offset, int W, int H)
{
    const int maxv = sizeof(T) == 1 ? 255 : 1023;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W * H; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const int c = src[y * stride + x];
        if (kind == 4) {
            const int bo = c >> (sizeof(T) == 1 ? 3 : 5);
            dst[y * stride + x] = (bo < band || bo > band + 3) ? (T)c : (T)f_clip3(0, maxv, c + offset[bo - band]);
            continue;
        }
        const int dx0 = (kind == 1) ? 0 : (kind == 3 ? 1 : -1), dy0 = (kind == 0) ? 0 : -1;
        int n[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int xx = x + (s ? -dx0 : dx0), yy = y + (s ? -dy0 : dy0);
            n[s] = (yy < 0) ? (int)upper[xx] : (xx < 0) ? (int)left[yy] : (int)src[yy * stride + xx];
        }
        dst[y * stride + x] = (T)f_clip3(0, maxv, c + offset[f_sgn(c, n[0]) + f_sgn(c, n[1]) + 2]);
    }
}

/* ---------------- SAO apply, whole picture (out of place) ---------------- */
/* ApplySaoOffsetsPicture(16bit) -> ApplySaoOffsetsLcu(16bit) (Codec/EbEncDecProcess.c:215-757, :762-1330): every sample is
 * classified against the UNFILTERED neighbours (the reference keeps the previous LCU's last column / the previous LCU
 * row's last row aside for exactly that), so src -> dst in one streaming pass.  One thread per 8 consecutive samples of
 * a row (one 8- or 16-byte store); the three planes are three slices of one launch. */
struct SaoLcuParams { uint8_t merge_left, merge_up, edge_flags, pad; uint32_t type[2]; int32_t offset[3][4]; uint32_t band[3]; }; /* = SvtAmdSaoLcuParams */
struct SaoPic {
    const void *src[3];
    void *dst[3];
    const SaoLcuParams *lcus;
    int strideY, strideC, width, height, lcuCols, lumaOn, chromaOn;
};

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void k_sao_apply_picture(const SaoPic P, uint32_t groupsY, uint32_t groupsC)
{
    constexpr int maxv = sizeof(T) == 1 ? 255 : 1023, boShift = sizeof(T) == 1 ? 3 : 5;
    typedef typename std::conditional<sizeof(T) == 1, uint2, uint4>::type V8; /* 8 samples */
    const uint32_t total = groupsY + 2 * groupsC;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += gridDim.x * blockDim.x) {
        const int comp = g < groupsY ? 0 : (g < groupsY + groupsC ? 1 : 2);
        const uint32_t gi = comp == 0 ? g : (comp == 1 ? g - groupsY : g - groupsY - groupsC);
        const int sh = comp ? 1 : 0, W = P.width >> sh, H = P.height >> sh, stride = comp ? P.strideC : P.strideY, L = 64 >> sh;
        const int gpr = (W + 7) >> 3, y = (int)(gi / gpr), x0 = (int)(gi - (uint32_t)y * gpr) * 8;
        const T *src = (const T *)P.src[comp];
        T *dst = (T *)P.dst[comp];
        const SaoLcuParams *p = P.lcus + (y / L) * P.lcuCols + (x0 / L);
        const uint32_t type = (comp ? P.chromaOn : P.lumaOn) ? p->type[comp ? 1 : 0] : 0u;
        const int n = W - x0 < 8 ? W - x0 : 8;
        const T *row = src + (size_t)y * stride;
        int c[10]; /* c[1..8] = the group, c[0] / c[9] = the samples left / right of it */
        if (VEC && n == 8) {
            const V8 v = *(const V8 *)(row + x0);
            const uint32_t *w = (const uint32_t *)&v;
#pragma unroll
            for (int k = 0; k < 8; k++)
                c[k + 1] = sizeof(T) == 1 ? (int)((w[k >> 2] >> (8 * (k & 3))) & 255u) : (int)((w[k >> 1] >> (16 * (k & 1))) & 65535u);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++)
                c[k + 1] = k < n ? (int)row[x0 + k] : 0;
        }
        int out[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            out[k] = c[k + 1];
        if (type == 5) {
            const int pos = (int)p->band[comp];
            const int o0 = (int8_t)p->offset[comp][0], o1 = (int8_t)p->offset[comp][1], o2 = (int8_t)p->offset[comp][2], o3 = (int8_t)p->offset[comp][3];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int b = (c[k + 1] >> boShift) - pos;
                const int o = b == 0 ? o0 : b == 1 ? o1 : b == 2 ? o2 : b == 3 ? o3 : 0;
                out[k] = f_clip3(0, maxv, c[k + 1] + o);
            }
        } else if (type >= 1 && type <= 4) {
            const int lx0 = x0 % L, ly = y % L, lcuX = x0 - lx0, lcuY = y - ly;
            const int lw = W - lcuX < L ? W - lcuX : L, lh = H - lcuY < L ? H - lcuY : L;
            const int ef = p->edge_flags;
            const bool rowSkip = type != 1 && ((ly == 0 && (ef & 4)) || (ly == lh - 1 && (ef & 8)));
            const int o[5] = {(int8_t)p->offset[comp][0], (int8_t)p->offset[comp][1], 0, (int8_t)p->offset[comp][2], (int8_t)p->offset[comp][3]};
            int a[8], b[8]; /* the two neighbours of every sample */
            if (type == 1) {
                c[0] = x0 > 0 ? (int)row[x0 - 1] : 0;
                c[9] = x0 + 8 < W ? (int)row[x0 + 8] : 0;
#pragma unroll
                for (int k = 0; k < 8; k++)
                    a[k] = c[k], b[k] = c[k + 2];
            } else {
                /* upper / lower rows at column offset -dx / +dx ... a = (y-1, x+dx), b = (y+1, x-dx) */
                const int dx = type == 2 ? 0 : (type == 4 ? 1 : -1);
                const T *up = y > 0 ? row - stride : row, *dn = y + 1 < H ? row + stride : row;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int xa = x0 + k + dx, xb = x0 + k - dx;
                    a[k] = (xa >= 0 && xa < W) ? (int)up[xa] : 0;
                    b[k] = (xb >= 0 && xb < W) ? (int)dn[xb] : 0;
                }
            }
            if (!rowSkip) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int lx = lx0 + k;
                    const bool colSkip = type != 2 && ((lx == 0 && (ef & 1)) || (lx == lw - 1 && (ef & 2)));
                    if (!colSkip)
                        out[k] = f_clip3(0, maxv, c[k + 1] + o[f_sgn(c[k + 1], a[k]) + f_sgn(c[k + 1], b[k]) + 2]);
                }
            }
        }
        T *drow = dst + (size_t)y * stride + x0;
        if (VEC && n == 8) {
            V8 v;
            uint32_t *w = (uint32_t *)&v;
            if (sizeof(T) == 1) {
                w[0] = (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16) | ((uint32_t)out[3] << 24);
                w[1] = (uint32_t)out[4] | ((uint32_t)out[5] << 8) | ((uint32_t)out[6] << 16) | ((uint32_t)out[7] << 24);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    w[k] = (uint32_t)out[2 * k] | ((uint32_t)out[2 * k + 1] << 16);
            }
            *(V8 *)drow = v;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < n)
                    drow[k] = (T)out[k];
        }
    }
}

/* ---------------- pack / unpack (streaming) ---------------- */
__global__ void k_pack(const uint8_t *__restrict__ in8, uint32_t in8Stride, const uint8_t *__restrict__ inn,
                       uint32_t innStride, uint16_t *__restrict__ out16, uint32_t outStride, uint32_t w, uint32_t h,
                       int compressed)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / w, x = i - y * w;
        const uint32_t two = compressed ? (inn[(x >> 2) + (size_t)y * innStride] >> (6 - 2 * (x & 3))) & 3u
                                        : (inn[x + (size_t)y * innStride] >> 6) & 3u;
        out16[x + (size_t)y * outStride] = (uint16_t)((in8[x + (size_t)y * in8Stride] << 2) | two);
    }
}
__global__ void k_cpack(const uint8_t *__restrict__ inn, uint32_t innStride, uint8_t *__restrict__ out,
                        uint32_t outStride, uint32_t w, uint32_t h)
{
    const uint32_t w4 = w / 4;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w4 * h; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / w4, x = (i - y * w4) * 4;
        const uint8_t *p = inn + x + (size_t)y * innStride;
        out[(x >> 2) + (size_t)y * outStride] =
            (uint8_t)((p[0] & 0xC0) | ((p[1] >> 2) & 0x30) | ((p[2] >> 4) & 0x0C) | ((p[3] >> 6) & 0x03));
    }
}
/* eight samples a thread where the three planes allow 16 / 8 / 8-byte accesses (`vec`, uniform: the encoder's picture buffers and the application's
 * 16-bit input do), a sample a thread otherwise; 2 B read + 2 B written per sample: HBM-bound */
__global__ void k_unpack(const uint16_t *__restrict__ in16, uint32_t inStride, uint8_t *__restrict__ out8,
                         uint32_t out8Stride, uint8_t *__restrict__ outn, uint32_t outnStride, uint32_t w, uint32_t h)
{
    const bool vec = !(((uintptr_t)in16 | ((uintptr_t)inStride << 1)) & 15) && !(((uintptr_t)out8 | out8Stride) & 7) &&
                     (!outn || !(((uintptr_t)outn | outnStride) & 7));
    if (vec) {
        const uint32_t gw = (w + 7) >> 3, full = w >> 3;
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < gw * h; i += gridDim.x * blockDim.x) {
            const uint32_t y = i / gw, g = i - y * gw, x = g << 3;
            if (g < full) {
                const uint4 v = *(const uint4 *)(in16 + x + (size_t)y * inStride);
                /* sample pairs (lo | hi << 16): the 8 MSBs are bits 2..9, the two LSBs go to bits 6..7 of their byte */
                const uint32_t p[4] = {v.x, v.y, v.z, v.w};
                uint32_t m[2], l[2];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t a = p[2 * k], b = p[2 * k + 1];
                    m[k] = ((a >> 2) & 255u) | (((a >> 18) & 255u) << 8) | (((b >> 2) & 255u) << 16) | (((b >> 18) & 255u) << 24);
                    l[k] = ((a & 3u) << 6) | (((a >> 16) & 3u) << 14) | ((b & 3u) << 22) | (((b >> 16) & 3u) << 30);
                }
                *(uint2 *)(out8 + x + (size_t)y * out8Stride) = make_uint2(m[0], m[1]);
                if (outn)
                    *(uint2 *)(outn + x + (size_t)y * outnStride) = make_uint2(l[0], l[1]);
            } else {
                for (uint32_t xx = x; xx < w; xx++) {
                    const uint16_t q = in16[xx + (size_t)y * inStride];
                    out8[xx + (size_t)y * out8Stride] = (uint8_t)(q >> 2);
                    if (outn)
                        outn[xx + (size_t)y * outnStride] = (uint8_t)(q << 6);
                }
            }
        }
        return;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / w, x = i - y * w;
        const uint16_t p = in16[x + (size_t)y * inStride];
        out8[x + (size_t)y * out8Stride] = (uint8_t)(p >> 2);
        if (outn)
            outn[x + (size_t)y * outnStride] = (uint8_t)(p << 6);
    }
}
__global__ void k_unpack_avg(const uint16_t *__restrict__ l0, uint32_t s0, const uint16_t *__restrict__ l1, uint32_t s1,
                             uint8_t *__restrict__ dst, uint32_t ds, uint32_t w, uint32_t h)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / w, x = i - y * w;
        dst[x + (size_t)y * ds] = (uint8_t)(((uint8_t)(l0[x + (size_t)y * s0] >> 2) + (uint8_t)(l1[x + (size_t)y * s1] >> 2) + 1) >> 1);
    }
}

/* ------------------------------------------------------------------------------------------------------------------------
 * SAO parameter decision from per-LCU statistics (SaoGenerationDecision(16bit) after its gathering step,
 * EbSampleAdaptiveOffsetGenerationDecision.c:44-640, :853-930).  Two launches:
 *   k_sao_decide_own   one thread per LCU: the LCU's own best luma / chroma parameters and their costs (independent)
 *   k_sao_decide_merge one workgroup: the merge-left / merge-up test needs the neighbours' FINAL parameters, so the LCUs are
 *                      visited as a wavefront over anti-diagonals (x + y = d), one thread per LCU row
 * ------------------------------------------------------------------------------------------------------------------------ */
struct SaoDecide {                                  /* = SvtAmdSaoDecisionParams */
    uint64_t lambda, chromaLambda;
    uint32_t typeBits[6], mergeBits[2], offsetBits[8];
    uint8_t is10, mmSao, temporalLayer, pad;
};
__device__ __forceinline__ int64_t sao_rate_cost(uint64_t rate, uint64_t lambda) { return (int64_t)((rate * lambda + (1u << 22)) >> 23); }
__device__ __forceinline__ int sao_est(int diff, int count, int lo, int hi)
{
    const int o = count == 0 ? 0 : diff / count;
    return o < lo ? lo : o > hi ? hi : o;
}
__device__ __forceinline__ int sao_dist(int o, int diff, int count) { return -(2 * o * diff) + count * o * o; }
__device__ __forceinline__ uint32_t sao_offset_bits(const SaoDecide &P, int o)
{
    const int a = o < 0 ? -o : o;
    return P.offsetBits[a > 7 ? 7 : a];
}
/* best edge-offset class of one component set: comps = 1 (luma) or 2 (Cb + Cr, the distortion keeps accumulating across the
 * two components exactly as the reference's eoTypeDistortion does) */
template <int COMPS>
__device__ __forceinline__ int64_t sao_best_eo(const SaoDecide &P, const SaoStats *const *S, int firstType, uint64_t lambda, int sh,
                                              int m, uint32_t &bestType)
{
    int64_t best = (int64_t)(~0ull >> 1);
    bestType = 0;
    for (int t = firstType; t < 4; t++) {
        int64_t d = 0, cost = 0;
        for (int c = 0; c < COMPS; c++) {
            uint64_t bits = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int diff = S[c]->eoDiff[t][k], cnt = S[c]->eoCount[t][k];
                const int o = sao_est(diff, cnt, k < 2 ? 0 : -m, k < 2 ? m : 0);
                d += sao_dist(o, diff, cnt) >> sh;
                bits += sao_offset_bits(P, o);
            }
            cost += (d << 8) + sao_rate_cost(bits, lambda);
        }
        if (cost < best)
            best = cost, bestType = (uint32_t)t;
    }
    return best + sao_rate_cost(P.typeBits[bestType + 1], lambda);
}
__device__ __forceinline__ void sao_eo_offsets(const SaoStats *S, uint32_t t, int m, int32_t *o)
{
#pragma unroll
    for (int k = 0; k < 4; k++)
        o[k] = sao_est(S->eoDiff[t][k], S->eoCount[t][k], k < 2 ? 0 : -m, k < 2 ? m : 0);
}

__global__ void __launch_bounds__(64) k_sao_decide_own(SaoDecide P, const SaoStats *sy, const SaoStats *scb, const SaoStats *scr,
                                                       uint32_t nlcu, const uint8_t *enable, SaoLcuParams *params, int64_t *costs)
{
    /* the three statistics records of the workgroup's 64 LCUs, fetched with coalesced loads (a thread walking its own 312-byte
     * records would pay one memory latency per band) */
    __shared__ SaoStats st[3][64];
    {
        const uint32_t first = blockIdx.x * 64, cnt = min(64u, nlcu - first);
        const uint32_t words = cnt * (uint32_t)(sizeof(SaoStats) / 4);
        const SaoStats *src[3] = {sy, scb, scr};
        for (int c = 0; c < (P.mmSao ? 3 : 1); c++) {
            const uint32_t *g = (const uint32_t *)(src[c] + first);
            uint32_t *l = (uint32_t *)&st[c][0];
            for (uint32_t w = threadIdx.x; w < words; w += 64)
                l[w] = g[w];
        }
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nlcu || (enable != nullptr && enable[i] == 2)) /* 2: parameters given, a merge candidate only */
        return;
    SaoLcuParams o;
    o.merge_left = o.merge_up = 0, o.edge_flags = params[i].edge_flags, o.pad = 0;
    o.type[0] = o.type[1] = 0;
    for (int c = 0; c < 3; c++) {
        o.band[c] = 0;
        for (int k = 0; k < 4; k++)
            o.offset[c][k] = 0;
    }
    int64_t lumaBest = 0, chromaBest = 0;
    const bool reduced = !P.mmSao;
    if ((enable == nullptr || enable[i] == 1) && (P.mmSao || P.temporalLayer < 2)) {
        const int sh = P.is10 ? 4 : 0, m = P.is10 ? 31 : 7;
        const int64_t maxc = (int64_t)(~0ull >> 1);
        const SaoStats *Y = &st[0][threadIdx.x];
        {   /* luma */
            const int64_t offCost = sao_rate_cost(P.typeBits[0], P.lambda);
            int64_t boBest = maxc;
            uint32_t bestBand = 0;
            if (!reduced && !P.is10) {
                /* sliding window of four bands: distortion and rate of the last four */
                int64_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
                uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
                for (int b = 0; b < 32; b++) {
                    const int diff = Y->boDiff[b], cnt = Y->boCount[b];
                    const int ob = sao_est(diff, cnt, -m, m);
                    d0 = d1, d1 = d2, d2 = d3, d3 = sao_dist(ob, diff, cnt) >> sh;
                    r0 = r1, r1 = r2, r2 = r3, r3 = sao_offset_bits(P, ob) + (ob ? 32768u : 0u);
                    if (b >= 3) {
                        const int64_t c = ((d0 + d1 + d2 + d3) << 8) + sao_rate_cost(163840ull + r0 + r1 + r2 + r3, P.lambda);
                        if (c < boBest)
                            boBest = c, bestBand = (uint32_t)(b - 3);
                    }
                }
                boBest += sao_rate_cost(P.typeBits[5], P.lambda);
            }
            uint32_t bestEo;
            const SaoStats *one[1] = {Y};
            const int64_t eoBest = sao_best_eo<1>(P, one, reduced ? 1 : 0, P.lambda, sh, m, bestEo);
            if (boBest < offCost || eoBest < offCost) {
                if (boBest <= eoBest) {
                    lumaBest = boBest, o.type[0] = 5, o.band[0] = bestBand;
                    for (int k = 0; k < 4; k++)
                        o.offset[0][k] = sao_est(Y->boDiff[bestBand + k], Y->boCount[bestBand + k], -m, m);
                } else {
                    lumaBest = eoBest, o.type[0] = bestEo + 1;
                    sao_eo_offsets(Y, bestEo, m, o.offset[0]);
                }
            } else {
                lumaBest = offCost;
            }
        }
        if (P.mmSao) { /* chroma: edge offset only, Cb and Cr share the class */
            const int64_t offCost = sao_rate_cost(P.typeBits[0], P.chromaLambda);
            const SaoStats *two[2] = {&st[1][threadIdx.x], &st[2][threadIdx.x]};
            uint32_t bestEo;
            const int64_t eoBest = sao_best_eo<2>(P, two, 0, P.chromaLambda, sh, m, bestEo);
            if (eoBest < offCost) {
                chromaBest = eoBest, o.type[1] = bestEo + 1;
                sao_eo_offsets(two[0], bestEo, m, o.offset[1]);
                sao_eo_offsets(two[1], bestEo, m, o.offset[2]);
            } else {
                chromaBest = offCost;
            }
        }
    }
    params[i] = o;
    costs[2 * i] = lumaBest, costs[2 * i + 1] = chromaBest;
}

/* distortion of applying a neighbour's parameters to this LCU's statistics (TestSaoCopyModes :487-587) */
__device__ __forceinline__ int64_t sao_merge_dist(const SaoLcuParams &N, int comp, const SaoStats *S)
{
    const uint32_t type = N.type[comp ? 1 : 0];
    int64_t d = 0;
    if (type == 0)
        return 0;
    for (int k = 0; k < 4; k++) {
        const int o = N.offset[comp][k];
        if (type == 5) {
            const uint32_t b = (N.band[comp] + k) & 31;
            d += sao_dist(o, S->boDiff[b], S->boCount[b]);
        } else {
            d += sao_dist(o, S->eoDiff[type - 1][k], S->eoCount[type - 1][k]);
        }
    }
    return d;
}
/* What a neighbour hands on: 20 bytes instead of the 72-byte record */
struct SaoMergeCand { uint8_t type[2]; uint8_t band[3]; int8_t offset[3][4]; uint8_t pad[3]; };
/* the eight statistics a candidate needs for one component, fetched without a branch so that the loads of all components and
 * of both candidates are in flight together (a chain of "if type ... load ... use" pays one memory latency per link) */
struct SaoMergeFetch { int diff[4], cnt[4]; };
__device__ __forceinline__ SaoMergeFetch sao_merge_fetch(const SaoMergeCand &N, int comp, const SaoStats *S)
{
    const uint32_t type = N.type[comp ? 1 : 0];
    const uint32_t band = N.band[comp] > 28 ? 28u : N.band[comp];
    const int32_t *dp = type == 5 ? &S->boDiff[band] : &S->eoDiff[type ? type - 1 : 0][0];
    const uint16_t *cp = type == 5 ? &S->boCount[band] : &S->eoCount[type ? type - 1 : 0][0];
    SaoMergeFetch f;
#pragma unroll
    for (int k = 0; k < 4; k++)
        f.diff[k] = dp[k], f.cnt[k] = cp[k];
    return f;
}
__device__ __forceinline__ int64_t sao_merge_dist_f(const SaoMergeCand &N, int comp, const SaoMergeFetch &f)
{
    int64_t d = 0;
    if (N.type[comp ? 1 : 0] == 0)
        return 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        d += sao_dist((int)N.offset[comp][k], f.diff[k], f.cnt[k]);
    return d;
}
/* One workgroup; eight lanes walk LCU row y together: at step d they finish LCU (d - y, y).  Both merge candidates of that LCU
 * were finished at step d - 1 - the left one by the same lanes, the upper one by the lanes of row y - 1 - so the final
 * parameters travel through two LDS lines (previous / current anti-diagonal, indexed by row) and HBM only sees each LCU's own
 * record once in and once out.  Lanes 0..2 of a row price the left candidate's three components, lanes 3..5 the upper one's
 * (their eight statistics come from wherever the candidate's type and band point: one HBM round trip per step, which is what
 * a step costs - about 2 us, profiles/r01_m_leaf_kernel_stats.txt), lane 0 decides; it also fetches the next LCU's record
 * one step ahead.  MAXR rows fit the LDS lines; taller pictures take the generic kernel below. */
template <int MAXR>
__global__ void __launch_bounds__(MAXR * 8) k_sao_decide_merge_lds(SaoDecide P, const SaoStats *sy, const SaoStats *scb, const SaoStats *scr,
                                                                   uint32_t cols, uint32_t rows, const uint8_t *enable, SaoLcuParams *params,
                                                                   int64_t *costs)
{
    __shared__ SaoMergeCand line[2][MAXR];
    const int sh = P.is10 ? 4 : 0;
    const int64_t maxc = (int64_t)(~0ull >> 1);
    const uint32_t y = threadIdx.x >> 3, l = threadIdx.x & 7, lead = (threadIdx.x & 63) & ~7u; /* lane of the row's lane 0 inside the wave */
    const int64_t rMerge = sao_rate_cost(P.mergeBits[1], P.lambda);
    SaoLcuParams nxt;
    int64_t nLuma = 0, nChroma = 0;
    uint32_t nEn = 0;
    if (l == 0 && y == 0 && rows)
        nxt = params[0], nLuma = costs[0], nChroma = costs[1], nEn = enable ? enable[0] : 1;
    for (uint32_t d = 0; d < cols + rows - 1; d++) {
        const uint32_t x = d - y;
        const bool mine = y < rows && y <= d && x < cols;
        const uint32_t i = mine ? y * cols + x : 0;
        SaoLcuParams o;
        int64_t luma = 0, chroma = 0;
        uint32_t en = 0;
        if (l == 0) {
            o = nxt, luma = nLuma, chroma = nChroma, en = nEn;
            const uint32_t xn = d + 1 - y; /* prefetch for step d + 1: LCU (x + 1, y), or the row's first one */
            if (y < rows && y <= d + 1 && xn < cols) {
                const uint32_t in = y * cols + xn;
                nxt = params[in], nLuma = costs[2 * in], nChroma = costs[2 * in + 1], nEn = enable ? enable[in] : 1;
            }
        }
        const uint32_t flagsEn = (uint32_t)__shfl((int)(l == 0 ? (en | ((uint32_t)o.edge_flags << 8)) : 0u), (int)lead);
        const bool active = mine && (flagsEn & 0xff) == 1;
        const uint32_t edge = flagsEn >> 8;
        const bool hasLeft = !(edge & 1) && x > 0, hasUp = !(edge & 4) && y > 0;
        /* lanes 0..5: one (candidate, component) each */
        int64_t dist = 0;
        SaoMergeCand N;
        {
            const bool up = l >= 3;
            N = line[(d + 1) & 1][up ? (y ? y - 1 : 0) : y];
            if (!(up ? hasUp : hasLeft))
                N.type[0] = N.type[1] = 0;
            const int comp = up ? (int)l - 3 : (int)l;
            if (active && l < 6) {
                const SaoStats *S = (comp == 0 ? sy : comp == 1 ? scb : scr) + i;
                const SaoMergeFetch f = sao_merge_fetch(N, comp, S);
                dist = sao_merge_dist_f(N, comp, f);
            }
        }
        /* gather the six distortions in lane 0 of the row */
        const int64_t d1 = __shfl_down(dist, 1), d2 = __shfl_down(dist, 2), d3 = __shfl_down(dist, 3), d4 = __shfl_down(dist, 4),
                      d5 = __shfl_down(dist, 5);
        if (mine && l == 0) {
            if (active) {
                const uint64_t leftFlag = hasLeft ? P.mergeBits[0] : 0, upFlag = hasUp ? P.mergeBits[0] : 0;
                const int64_t flags = sao_rate_cost(leftFlag + upFlag, P.lambda);
                const int64_t best = luma + chroma + flags;
                luma += flags, chroma += flags;
                int64_t lCost = maxc, uCost = maxc, lLuma = 0, lChroma = 0, uLuma = 0, uChroma = 0;
                if (hasLeft) {
                    const int64_t dl = dist >> sh, dc = (d1 + d2) >> sh;
                    lLuma = (dl << 8) + rMerge, lChroma = (dc << 8) + rMerge, lCost = (dl << 8) + (dc << 8) + rMerge;
                }
                if (hasUp) {
                    const int64_t dl = d3 >> sh, dc = (d4 + d5) >> sh;
                    const int64_t r = sao_rate_cost(leftFlag + P.mergeBits[1], P.lambda);
                    uLuma = (dl << 8) + r, uChroma = (dc << 8) + r, uCost = (dl << 8) + (dc << 8) + r;
                }
                if (lCost < best || uCost < best) {
                    const bool left = lCost <= uCost && hasLeft;
                    if (left || hasUp) {
                        const SaoMergeCand M = line[(d + 1) & 1][left ? y : y - 1];
                        o.merge_left = left, o.merge_up = !left;
                        luma = left ? lLuma : uLuma, chroma = left ? lChroma : uChroma;
                        o.type[0] = M.type[0], o.type[1] = M.type[1];
                        for (int c = 0; c < 3; c++) {
                            o.band[c] = M.band[c];
                            for (int k = 0; k < 4; k++)
                                o.offset[c][k] = M.offset[c][k];
                        }
                        params[i] = o;
                    }
                }
                costs[2 * i] = luma, costs[2 * i + 1] = chroma;
            }
            SaoMergeCand fin;
            fin.type[0] = (uint8_t)o.type[0], fin.type[1] = (uint8_t)o.type[1];
            for (int c = 0; c < 3; c++) {
                fin.band[c] = (uint8_t)o.band[c];
                for (int k = 0; k < 4; k++)
                    fin.offset[c][k] = (int8_t)o.offset[c][k];
            }
            line[d & 1][y] = fin;
        }
        /* only the LDS lines travel between threads: wait for them, not for the HBM stores and the prefetch (a full
         * __syncthreads() would put a store acknowledgement on every step's critical path) */
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

__global__ void __launch_bounds__(256) k_sao_decide_merge(SaoDecide P, const SaoStats *sy, const SaoStats *scb, const SaoStats *scr,
                                                          uint32_t cols, uint32_t rows, const uint8_t *enable, SaoLcuParams *params,
                                                          int64_t *costs)
{
    const int sh = P.is10 ? 4 : 0;
    const int64_t maxc = (int64_t)(~0ull >> 1);
    for (uint32_t d = 0; d < cols + rows - 1; d++) {
        for (uint32_t y = threadIdx.x; y < rows; y += blockDim.x) {
            const uint32_t x = d - y;
            if (y > d || x >= cols)
                continue;
            const uint32_t i = y * cols + x;
            if (enable != nullptr && enable[i] != 1)
                continue;
            SaoLcuParams o = params[i];
            const bool hasLeft = !(o.edge_flags & 1) && x > 0, hasUp = !(o.edge_flags & 4) && y > 0;
            const uint64_t leftFlag = hasLeft ? P.mergeBits[0] : 0, upFlag = hasUp ? P.mergeBits[0] : 0;
            const int64_t flags = sao_rate_cost(leftFlag + upFlag, P.lambda);
            int64_t luma = costs[2 * i] + flags, chroma = costs[2 * i + 1] + flags;
            int64_t best = costs[2 * i] + costs[2 * i + 1] + flags;
            int64_t lCost = maxc, uCost = maxc, lLuma = 0, lChroma = 0, uLuma = 0, uChroma = 0;
            SaoLcuParams L, U;
            if (hasLeft) {
                L = params[i - 1];
                const int64_t dl = sao_merge_dist(L, 0, sy + i) >> sh,
                              dc = (sao_merge_dist(L, 1, scb + i) + sao_merge_dist(L, 2, scr + i)) >> sh;
                const int64_t r = sao_rate_cost(P.mergeBits[1], P.lambda);
                lLuma = (dl << 8) + r, lChroma = (dc << 8) + r, lCost = (dl << 8) + (dc << 8) + r;
            }
            if (hasUp) {
                U = params[i - cols];
                const int64_t dl = sao_merge_dist(U, 0, sy + i) >> sh,
                              dc = (sao_merge_dist(U, 1, scb + i) + sao_merge_dist(U, 2, scr + i)) >> sh;
                const int64_t r = sao_rate_cost(leftFlag + P.mergeBits[1], P.lambda);
                uLuma = (dl << 8) + r, uChroma = (dc << 8) + r, uCost = (dl << 8) + (dc << 8) + r;
            }
            if (lCost < best || uCost < best) {
                const bool left = lCost <= uCost && hasLeft;
                const SaoLcuParams &N = left ? L : U;
                if (left || hasUp) {
                    o.merge_left = left, o.merge_up = !left;
                    luma = left ? lLuma : uLuma, chroma = left ? lChroma : uChroma;
                    o.type[0] = N.type[0], o.type[1] = N.type[1];
                    for (int c = 0; c < 3; c++) {
                        o.band[c] = N.band[c];
                        for (int k = 0; k < 4; k++)
                            o.offset[c][k] = N.offset[c][k];
                    }
                    params[i] = o;
                }
            }
            costs[2 * i] = luma, costs[2 * i + 1] = chroma;
        }
        __threadfence();
        __syncthreads();
    }
}

static inline dim3 grid1d(uint32_t n) { return dim3((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

/* ---------------- batched C-ABI (device pointers) ---------------- */
extern "C" int svt_amd_dlf_luma_edges_batch(SvtAmdContext *ctx, void *d_plane, uint32_t stride, int bytes_per_sample,
                                            const SvtAmdDlfLumaEdge *d_edges, uint32_t nedges)
{
    if (!ctx || !d_plane || !d_edges || !nedges || (bytes_per_sample != 1 && bytes_per_sample != 2))
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    if (bytes_per_sample == 1)
        hipLaunchKernelGGL(k_dlf_luma<uint8_t>, grid1d(nedges), dim3(256), 0, ctx->stream, (uint8_t *)d_plane, (int)stride,
                           (const DlfLumaEdge *)d_edges, nedges);
    else
        hipLaunchKernelGGL(k_dlf_luma<uint16_t>, grid1d(nedges), dim3(256), 0, ctx->stream, (uint16_t *)d_plane, (int)stride,
                           (const DlfLumaEdge *)d_edges, nedges);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
extern "C" int svt_amd_dlf_chroma_edges_batch(SvtAmdContext *ctx, void *d_cb, void *d_cr, uint32_t stride,
                                              int bytes_per_sample, const SvtAmdDlfChromaEdge *d_edges, uint32_t nedges)
{
    if (!ctx || !d_cb || !d_cr || !d_edges || !nedges || (bytes_per_sample != 1 && bytes_per_sample != 2))
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    if (bytes_per_sample == 1)
        hipLaunchKernelGGL(k_dlf_chroma<uint8_t>, grid1d(nedges), dim3(256), 0, ctx->stream, (uint8_t *)d_cb, (uint8_t *)d_cr,
                           (int)stride, (const DlfChromaEdge *)d_edges, nedges);
    else
        hipLaunchKernelGGL(k_dlf_chroma<uint16_t>, grid1d(nedges), dim3(256), 0, ctx->stream, (uint16_t *)d_cb, (uint16_t *)d_cr,
                           (int)stride, (const DlfChromaEdge *)d_edges, nedges);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
extern "C" int svt_amd_dlf_picture(SvtAmdContext *ctx, int bytes_per_sample, void *d_y, uint32_t strideY, void *d_cb,
                                   void *d_cr, uint32_t strideC, uint32_t width, uint32_t height, const uint8_t *d_bs_v,
                                   const uint8_t *d_bs_h, const uint8_t *d_qp, uint32_t qpStride, int32_t tcOffset,
                                   int32_t betaOffset, int32_t cbQpOffset, int32_t crQpOffset)
{
    if (!ctx || !d_y || !d_cb || !d_cr || !d_bs_v || !d_bs_h || !d_qp || (bytes_per_sample != 1 && bytes_per_sample != 2) ||
        width < 8 || height < 8 || (width & 7) || (height & 7) || strideY < width || strideC < width / 2 ||
        qpStride < width / 8)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    DlfPic P = {d_y, d_cb, d_cr, d_bs_v, d_bs_h, d_qp, (int)strideY, (int)strideC, (int)width, (int)height, (int)qpStride,
                (int)((width + 63) >> 6), tcOffset, betaOffset, cbQpOffset, crQpOffset};
    const uint32_t cw = width / 2, ch = height / 2;
    for (int dir = 0; dir < 2; dir++) {
        /* segments along x, then totals: vertical edges at x = 8, 16, .. < W in 4-row (chroma 2-row) pieces;
         * horizontal edges at y = 8, 16, .. < H in 4-column (chroma 2-column) pieces */
        const uint32_t lx = dir ? width / 4 : (width - 1) / 8, ly = dir ? (height - 1) / 8 : height / 4;
        const uint32_t cx = dir ? cw / 2 : (cw - 1) / 8, cy = dir ? (ch - 1) / 8 : ch / 2;
        const uint32_t nL = lx * ly, nC = cx * cy;
        if (!(nL + nC))
            continue;
        if (bytes_per_sample == 1)
            hipLaunchKernelGGL(k_dlf_picture<uint8_t>, grid1d(nL + nC), dim3(256), 0, ctx->stream, P, dir, lx, nL, cx ? cx : 1, nC);
        else
            hipLaunchKernelGGL(k_dlf_picture<uint16_t>, grid1d(nL + nC), dim3(256), 0, ctx->stream, P, dir, lx, nL, cx ? cx : 1, nC);
    }
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
extern "C" int svt_amd_bs_picture(SvtAmdContext *ctx, const SvtAmdCuMapEntry *d_map, const uint8_t *d_cbf, uint32_t width, uint32_t height,
                                  int slice_type, uint64_t ref_poc0, uint64_t ref_poc1, const uint8_t *d_lcu_edge, uint8_t *d_bs_v,
                                  uint8_t *d_bs_h)
{
    static_assert(sizeof(CuMapEntry) == sizeof(SvtAmdCuMapEntry), "map entry layout");
    if (!ctx || !d_map || !d_cbf || !d_lcu_edge || !d_bs_v || !d_bs_h || width < 8 || height < 8 || (width & 7) || (height & 7) ||
        slice_type < 0 || slice_type > 3)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t n = 2 * (width / 8) * (height / 8), nlcu = ((width + 63) / 64) * ((height + 63) / 64);
    /* 4x4 positions off the 8x8 grid are never written by the reference either: the arrays start as zeros */
    HIP_TRY(hipMemsetAsync(d_bs_v, 0, (size_t)nlcu * 256, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_bs_h, 0, (size_t)nlcu * 256, ctx->stream));
    hipLaunchKernelGGL(k_bs_picture, grid1d(n), dim3(256), 0, ctx->stream, (const CuMapEntry *)d_map, d_cbf, (int)width, (int)height,
                       slice_type, (unsigned long long)ref_poc0, (unsigned long long)ref_poc1, d_lcu_edge, d_bs_v, d_bs_h);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
extern "C" int svt_amd_sao_apply_picture(SvtAmdContext *ctx, int bytes_per_sample, const void *const d_src[3], void *const d_dst[3],
                                         uint32_t strideY, uint32_t strideC, uint32_t width, uint32_t height,
                                         const SvtAmdSaoLcuParams *d_lcus, int luma_on, int chroma_on)
{
    if (!ctx || !d_src || !d_dst || !d_lcus || (bytes_per_sample != 1 && bytes_per_sample != 2) || !width || !height ||
        (width & 1) || (height & 1) || strideY < width || strideC < width / 2)
        return SVT_AMD_ERR_BAD_PARAM;
    for (int k = 0; k < 3; k++)
        if (!d_src[k] || !d_dst[k] || d_src[k] == d_dst[k])
            return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    SaoPic P;
    bool vec = (strideY % 8) == 0 && (strideC % 8) == 0;
    for (int k = 0; k < 3; k++) {
        P.src[k] = d_src[k], P.dst[k] = d_dst[k];
        vec = vec && ((uintptr_t)d_src[k] % (8 * bytes_per_sample)) == 0 && ((uintptr_t)d_dst[k] % (8 * bytes_per_sample)) == 0;
    }
    P.lcus = (const SaoLcuParams *)d_lcus;
    P.strideY = (int)strideY, P.strideC = (int)strideC, P.width = (int)width, P.height = (int)height;
    P.lcuCols = (int)((width + 63) >> 6), P.lumaOn = luma_on, P.chromaOn = chroma_on;
    const uint32_t gy = ((width + 7) / 8) * height, gc = ((width / 2 + 7) / 8) * (height / 2);
    const dim3 grid = grid1d(gy + 2 * gc);
    if (bytes_per_sample == 1) {
        if (vec)
            hipLaunchKernelGGL((k_sao_apply_picture<uint8_t, true>), grid, dim3(256), 0, ctx->stream, P, gy, gc);
        else
            hipLaunchKernelGGL((k_sao_apply_picture<uint8_t, false>), grid, dim3(256), 0, ctx->stream, P, gy, gc);
    } else {
        if (vec)
            hipLaunchKernelGGL((k_sao_apply_picture<uint16_t, true>), grid, dim3(256), 0, ctx->stream, P, gy, gc);
        else
            hipLaunchKernelGGL((k_sao_apply_picture<uint16_t, false>), grid, dim3(256), 0, ctx->stream, P, gy, gc);
    }
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
extern "C" int svt_amd_sao_decide_picture(SvtAmdContext *ctx, const SvtAmdSaoDecisionParams *params, const SvtAmdSaoStats *d_stats_y,
                                          const SvtAmdSaoStats *d_stats_cb, const SvtAmdSaoStats *d_stats_cr, uint32_t lcu_cols,
                                          uint32_t lcu_rows, const uint8_t *d_enable, SvtAmdSaoLcuParams *d_params, int64_t *d_costs)
{
    static_assert(sizeof(SaoDecide) == sizeof(SvtAmdSaoDecisionParams), "layout");
    if (!ctx || !params || !d_stats_y || !d_stats_cb || !d_stats_cr || !lcu_cols || !lcu_rows || !d_params || !d_costs)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    SaoDecide P;
    ::memcpy(&P, params, sizeof(P));
    const uint32_t nlcu = lcu_cols * lcu_rows;
    hipLaunchKernelGGL(k_sao_decide_own, dim3((nlcu + 63) / 64), dim3(64), 0, ctx->stream, P, (const SaoStats *)d_stats_y,
                       (const SaoStats *)d_stats_cb, (const SaoStats *)d_stats_cr, nlcu, d_enable, (SaoLcuParams *)d_params, d_costs);
    if (P.mmSao || P.temporalLayer < 2) {
        if (lcu_rows <= 64)
            hipLaunchKernelGGL(k_sao_decide_merge_lds<64>, dim3(1), dim3(512), 0, ctx->stream, P, (const SaoStats *)d_stats_y,
                               (const SaoStats *)d_stats_cb, (const SaoStats *)d_stats_cr, lcu_cols, lcu_rows, d_enable,
                               (SaoLcuParams *)d_params, d_costs);
        else if (lcu_rows <= 128)
            hipLaunchKernelGGL(k_sao_decide_merge_lds<128>, dim3(1), dim3(1024), 0, ctx->stream, P, (const SaoStats *)d_stats_y,
                               (const SaoStats *)d_stats_cb, (const SaoStats *)d_stats_cr, lcu_cols, lcu_rows, d_enable,
                               (SaoLcuParams *)d_params, d_costs);
        else
            hipLaunchKernelGGL(k_sao_decide_merge, dim3(1), dim3(256), 0, ctx->stream, P, (const SaoStats *)d_stats_y,
                               (const SaoStats *)d_stats_cb, (const SaoStats *)d_stats_cr, lcu_cols, lcu_rows, d_enable,
                               (SaoLcuParams *)d_params, d_costs);
    }
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
/* one LCU with its neighbours' final parameters given (host pointers): the call SaoGenerationDecision(16bit) makes after its
 * gathering step.  A 2x2 grid: (1,0) = the upper LCU, (0,1) = the left one (both "given"), (1,1) = this LCU. */
extern "C" int svt_amd_sao_decide_lcu(SvtAmdContext *ctx, const SvtAmdSaoDecisionParams *params, const SvtAmdSaoStats *stats_y,
                                      const SvtAmdSaoStats *stats_cb, const SvtAmdSaoStats *stats_cr, const SvtAmdSaoLcuParams *left,
                                      const SvtAmdSaoLcuParams *up, SvtAmdSaoLcuParams *out, int64_t costs[2])
{
    if (!ctx || !params || !stats_y || !stats_cb || !stats_cr || !out || !costs)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    struct { SaoStats st[3][4]; SaoLcuParams lcu[4]; int64_t cost[8]; uint8_t enable[8]; } h;
    ::memset(&h, 0, sizeof(h));
    ::memcpy(&h.st[0][3], stats_y, sizeof(SaoStats));
    ::memcpy(&h.st[1][3], stats_cb, sizeof(SaoStats));
    ::memcpy(&h.st[2][3], stats_cr, sizeof(SaoStats));
    h.enable[1] = h.enable[2] = 2, h.enable[3] = 1;
    if (up)
        ::memcpy(&h.lcu[1], up, sizeof(SaoLcuParams));
    if (left)
        ::memcpy(&h.lcu[2], left, sizeof(SaoLcuParams));
    h.lcu[3].edge_flags = (uint8_t)((left ? 0 : 1) | (up ? 0 : 4));
    DBuf d(&h, sizeof(h));
    if (!d.ok)
        return SVT_AMD_ERR_DEVICE;
    auto *dh = (decltype(h) *)d.d;
    const int rc = svt_amd_sao_decide_picture(ctx, params, (const SvtAmdSaoStats *)dh->st[0], (const SvtAmdSaoStats *)dh->st[1],
                                              (const SvtAmdSaoStats *)dh->st[2], 2, 2, dh->enable, (SvtAmdSaoLcuParams *)dh->lcu, dh->cost);
    if (rc != SVT_AMD_OK)
        return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (!d.download(&h, sizeof(h)))
        return SVT_AMD_ERR_DEVICE;
    const uint8_t keep = out->edge_flags;
    ::memcpy(out, &h.lcu[3], sizeof(SaoLcuParams));
    out->edge_flags = keep;
    costs[0] = h.cost[6], costs[1] = h.cost[7];
    return SVT_AMD_OK;
}
extern "C" int svt_amd_sao_gather_picture(SvtAmdContext *ctx, int bytes_per_sample, const void *d_input,
                                          uint32_t inputStride, const void *d_recon, uint32_t reconStride,
                                          uint32_t width, uint32_t height, uint32_t lcu_size, int only_eo_90_45_135,
                                          SvtAmdSaoStats *d_stats)
{
    if (!ctx || !d_input || !d_recon || !d_stats || !width || !height || lcu_size < 8 ||
        (bytes_per_sample != 1 && bytes_per_sample != 2))
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    const int lw = (int)((width + lcu_size - 1) / lcu_size), lh = (int)((height + lcu_size - 1) / lcu_size);
    if (bytes_per_sample == 1)
        hipLaunchKernelGGL(k_sao_gather<uint8_t>, dim3(lw * lh), dim3(256), 0, ctx->stream, (const uint8_t *)d_input,
                           (int)inputStride, (const uint8_t *)d_recon, (int)reconStride, (int)width, (int)height,
                           (int)lcu_size, lw, only_eo_90_45_135, (SaoStats *)d_stats);
    else
        hipLaunchKernelGGL(k_sao_gather<uint16_t>, dim3(lw * lh), dim3(256), 0, ctx->stream, (const uint16_t *)d_input,
                           (int)inputStride, (const uint16_t *)d_recon, (int)reconStride, (int)width, (int)height,
                           (int)lcu_size, lw, only_eo_90_45_135, (SaoStats *)d_stats);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
extern "C" int svt_amd_pack_plane(SvtAmdContext *ctx, const uint8_t *d_in8, uint32_t in8Stride, const uint8_t *d_inn,
                                  uint32_t innStride, int compressed, uint16_t *d_out16, uint32_t outStride,
                                  uint32_t width, uint32_t height)
{
    if (!ctx || !d_in8 || !d_inn || !d_out16 || !width || !height)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_pack, grid1d(width * height), dim3(256), 0, ctx->stream, d_in8, in8Stride, d_inn, innStride, d_out16,
                       outStride, width, height, compressed);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
extern "C" int svt_amd_unpack_plane(SvtAmdContext *ctx, const uint16_t *d_in16, uint32_t inStride, uint8_t *d_out8,
                                    uint32_t out8Stride, uint8_t *d_outn, uint32_t outnStride, uint32_t width,
                                    uint32_t height)
{
    if (!ctx || !d_in16 || !d_out8 || !width || !height)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_unpack, grid1d(width * height), dim3(256), 0, ctx->stream, d_in16, inStride, d_out8, out8Stride, d_outn,
                       outnStride, width, height);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* ---------------- LEAF wrappers (host pointers, reference signatures) ---------------- */
template <typename T>
static void luma_dlf_leaf(T *edge, uint32_t stride, uint8_t vertical, int32_t tc, int32_t beta)
{
    /* stage the 8 x 4 neighbourhood: rows/cols -4..3 around the edge start */
    const int fs = vertical ? 1 : (int)stride, ns = vertical ? (int)stride : 1;
    const ptrdiff_t lo = -4 * fs, hi = 3 * fs + 3 * ns;
    DBuf buf(edge + lo, (size_t)(hi - lo + 1) * sizeof(T)), e(nullptr, sizeof(DlfLumaEdge), false);
    if (!(buf.ok && e.ok))
        return;
    DlfLumaEdge he = {(int32_t)(-lo), (int16_t)tc, (int16_t)beta, vertical, {0, 0, 0}};
    if (hipMemcpy(e.d, &he, sizeof(he), hipMemcpyHostToDevice) != hipSuccess)
        return;
    hipLaunchKernelGGL(k_dlf_luma<T>, dim3(1), dim3(64), 0, 0, (T *)buf.d, (int)stride, (const DlfLumaEdge *)e.d, 1u);
    if (finish("Luma4SampleEdgeDLFCore"))
        buf.download(edge + lo, (size_t)(hi - lo + 1) * sizeof(T));
}
extern "C" void svt_amd_Luma4SampleEdgeDLFCore(uint8_t *edgeStartFilteredSamplePtr, uint32_t reconLumaPicStride,
                                               uint8_t isVerticalEdge, int32_t tc, int32_t beta)
{
    luma_dlf_leaf<uint8_t>(edgeStartFilteredSamplePtr, reconLumaPicStride, isVerticalEdge, tc, beta);
}
extern "C" void svt_amd_Luma4SampleEdgeDLFCore16bit(uint16_t *edgeStartFilteredSamplePtr, uint32_t reconLumaPicStride,
                                                    uint8_t isVerticalEdge, int32_t tc, int32_t beta)
{
    luma_dlf_leaf<uint16_t>(edgeStartFilteredSamplePtr, reconLumaPicStride, isVerticalEdge, tc, beta);
}
template <typename T>
static void chroma_dlf_leaf(T *cb, T *cr, uint32_t stride, uint8_t vertical, uint8_t cbTc, uint8_t crTc)
{
    const int fs = vertical ? 1 : (int)stride, ns = vertical ? (int)stride : 1;
    const ptrdiff_t lo = -2 * fs, hi = fs + ns;
    DBuf b(cb + lo, (size_t)(hi - lo + 1) * sizeof(T)), r(cr + lo, (size_t)(hi - lo + 1) * sizeof(T)),
        e(nullptr, sizeof(DlfChromaEdge), false);
    if (!(b.ok && r.ok && e.ok))
        return;
    DlfChromaEdge he = {(int32_t)(-lo), cbTc, crTc, vertical, 0};
    if (hipMemcpy(e.d, &he, sizeof(he), hipMemcpyHostToDevice) != hipSuccess)
        return;
    hipLaunchKernelGGL(k_dlf_chroma<T>, dim3(1), dim3(64), 0, 0, (T *)b.d, (T *)r.d, (int)stride, (const DlfChromaEdge *)e.d, 1u);
    if (finish("Chroma2SampleEdgeDLFCore")) {
        b.download(cb + lo, (size_t)(hi - lo + 1) * sizeof(T));
        r.download(cr + lo, (size_t)(hi - lo + 1) * sizeof(T));
    }
}
extern "C" void svt_amd_Chroma2SampleEdgeDLFCore(uint8_t *edgeStartSampleCb, uint8_t *edgeStartSampleCr,
                                                 uint32_t reconChromaPicStride, uint8_t isVerticalEdge, uint8_t cbTc,
                                                 uint8_t crTc)
{
    chroma_dlf_leaf<uint8_t>(edgeStartSampleCb, edgeStartSampleCr, reconChromaPicStride, isVerticalEdge, cbTc, crTc);
}
extern "C" void svt_amd_Chroma2SampleEdgeDLFCore16bit(uint16_t *edgeStartSampleCb, uint16_t *edgeStartSampleCr,
                                                      uint32_t reconChromaPicStride, uint8_t isVerticalEdge,
                                                      uint8_t cbTc, uint8_t crTc)
{
    chroma_dlf_leaf<uint16_t>(edgeStartSampleCb, edgeStartSampleCr, reconChromaPicStride, isVerticalEdge, cbTc, crTc);
}

template <typename T>
static int sao_gather_leaf(int only_eo, T *input, uint32_t inputStride, T *recon, uint32_t reconStride, uint32_t lcuWidth,
                           uint32_t lcuHeight, int32_t *boDiff, uint16_t *boCount, int32_t eoDiff[4][5], uint16_t eoCount[4][5])
{
    DBuf a(input, span(inputStride, lcuWidth, lcuHeight) * sizeof(T)), b(recon, span(reconStride, lcuWidth, lcuHeight) * sizeof(T)),
        o(nullptr, sizeof(SaoStats), false);
    if (!(a.ok && b.ok && o.ok))
        return 1;
    hipLaunchKernelGGL(k_sao_gather<T>, dim3(1), dim3(256), 0, 0, (const T *)a.d, (int)inputStride, (const T *)b.d,
                       (int)reconStride, (int)lcuWidth, (int)lcuHeight, 1 << 20, 1, only_eo, (SaoStats *)o.d);
    SaoStats st;
    if (!finish("GatherSaoStatistics") || !o.download(&st, sizeof(st)))
        return 1;
    if (!only_eo) {
        ::memcpy(boDiff, st.boDiff, sizeof(st.boDiff));
        ::memcpy(boCount, st.boCount, sizeof(st.boCount));
    }
    ::memcpy(eoDiff, st.eoDiff, sizeof(st.eoDiff));
    ::memcpy(eoCount, st.eoCount, sizeof(st.eoCount));
    if (only_eo) /* the reference zeroes type 0 and never touches it */
        for (int k = 0; k < 5; k++)
            eoDiff[0][k] = 0, eoCount[0][k] = 0;
    return 0;
}
extern "C" int svt_amd_GatherSaoStatisticsLcuLossy_62x62(uint8_t *inputSamplePtr, uint32_t inputStride, uint8_t *reconSamplePtr,
                                                         uint32_t reconStride, uint32_t lcuWidth, uint32_t lcuHeight,
                                                         int32_t *boDiff, uint16_t *boCount, int32_t eoDiff[4][5],
                                                         uint16_t eoCount[4][5])
{
    return sao_gather_leaf<uint8_t>(0, inputSamplePtr, inputStride, reconSamplePtr, reconStride, lcuWidth, lcuHeight, boDiff, boCount, eoDiff, eoCount);
}
extern "C" int svt_amd_GatherSaoStatisticsLcu_62x62_16bit(uint16_t *inputSamplePtr, uint32_t inputStride, uint16_t *reconSamplePtr,
                                                          uint32_t reconStride, uint32_t lcuWidth, uint32_t lcuHeight,
                                                          int32_t *boDiff, uint16_t *boCount, int32_t eoDiff[4][5],
                                                          uint16_t eoCount[4][5])
{
    return sao_gather_leaf<uint16_t>(0, inputSamplePtr, inputStride, reconSamplePtr, reconStride, lcuWidth, lcuHeight, boDiff, boCount, eoDiff, eoCount);
}
extern "C" int svt_amd_GatherSaoStatisticsLcu_OnlyEo_90_45_135_Lossy(uint8_t *inputSamplePtr, uint32_t inputStride,
                                                                     uint8_t *reconSamplePtr, uint32_t reconStride,
                                                                     uint32_t lcuWidth, uint32_t lcuHeight,
                                                                     int32_t eoDiff[4][5], uint16_t eoCount[4][5])
{
    return sao_gather_leaf<uint8_t>(1, inputSamplePtr, inputStride, reconSamplePtr, reconStride, lcuWidth, lcuHeight, nullptr, nullptr, eoDiff, eoCount);
}
extern "C" int svt_amd_GatherSaoStatisticsLcu_62x62_OnlyEo_90_45_135_16bit(uint16_t *inputSamplePtr, uint32_t inputStride,
                                                                           uint16_t *reconSamplePtr, uint32_t reconStride,
                                                                           uint32_t lcuWidth, uint32_t lcuHeight,
                                                                           int32_t eoDiff[4][5], uint16_t eoCount[4][5])
{
    return sao_gather_leaf<uint16_t>(1, inputSamplePtr, inputStride, reconSamplePtr, reconStride, lcuWidth, lcuHeight, nullptr, nullptr, eoDiff, eoCount);
}

template <typename T>
static int sao_apply_leaf(int kind, T *recon, uint32_t stride, T *left, T *upper, uint32_t band, int8_t *offset,
                          uint32_t lcuHeight, uint32_t lcuWidth)
{
    /* the source copy includes the right column / bottom row neighbours the EO type reads */
    const uint32_t ex = (kind == 0 || kind == 2 || kind == 3), ey = (kind >= 1 && kind <= 3);
    const size_t bytes = span(stride, lcuWidth + ex, lcuHeight + ey) * sizeof(T);
    /* left[0..H-1] (+1 for EO_45), upper[-1..W] for the diagonals, upper[0..W-1] for EO_90 */
    const uint32_t nl = lcuHeight + (kind == 3), ulo = (kind == 2 || kind == 3) ? 1 : 0,
                   nu = lcuWidth + ((kind == 2 || kind == 3) ? 2 : 0);
    DBuf s(recon, bytes), d(recon, bytes), l(left ? (const void *)left : (const void *)recon, (left ? nl : 1) * sizeof(T)),
        u(upper ? (const void *)(upper - ulo) : (const void *)recon, (upper ? nu : 1) * sizeof(T)), o(offset, kind == 4 ? 4 : 5);
    if (!(s.ok && d.ok && l.ok && u.ok && o.ok))
        return 1;
    hipLaunchKernelGGL(k_sao_apply<T>, dim3(16), dim3(256), 0, 0, kind, (const T *)s.d, (T *)d.d, (int)stride, (const T *)l.d,
                       (const T *)u.d + ulo, (int)band, (const int8_t *)o.d, (int)lcuWidth, (int)lcuHeight);
    if (!finish("SAOApply"))
        return 1;
    /* copy back only the LCU area */
    T *tmp = (T *)malloc(bytes);
    if (!tmp || !d.download(tmp, bytes)) {
        free(tmp);
        return 1;
    }
    for (uint32_t y = 0; y < lcuHeight; y++)
        ::memcpy(recon + (size_t)y * stride, tmp + (size_t)y * stride, lcuWidth * sizeof(T));
    free(tmp);
    return 0;
}
#define SAO_LEAF(T, sfx)                                                                                                  \
    extern "C" int svt_amd_SAOApplyBO##sfx(T *r, uint32_t st, uint32_t band, int8_t *off, uint32_t h, uint32_t w)          \
    { return sao_apply_leaf<T>(4, r, st, nullptr, nullptr, band, off, h, w); }                                            \
    extern "C" int svt_amd_SAOApplyEO_0##sfx(T *r, uint32_t st, T *left, int8_t *off, uint32_t h, uint32_t w)              \
    { return sao_apply_leaf<T>(0, r, st, left, nullptr, 0, off, h, w); }                                                  \
    extern "C" int svt_amd_SAOApplyEO_90##sfx(T *r, uint32_t st, T *upper, int8_t *off, uint32_t h, uint32_t w)            \
    { return sao_apply_leaf<T>(1, r, st, nullptr, upper, 0, off, h, w); }                                                 \
    extern "C" int svt_amd_SAOApplyEO_135##sfx(T *r, uint32_t st, T *left, T *upper, int8_t *off, uint32_t h, uint32_t w)  \
    { return sao_apply_leaf<T>(2, r, st, left, upper, 0, off, h, w); }                                                    \
    extern "C" int svt_amd_SAOApplyEO_45##sfx(T *r, uint32_t st, T *left, T *upper, int8_t *off, uint32_t h, uint32_t w)   \
    { return sao_apply_leaf<T>(3, r, st, left, upper, 0, off, h, w); }
SAO_LEAF(uint8_t, )
extern "C" int svt_amd_SAOApplyBO16bit(uint16_t *r, uint32_t st, uint32_t band, int8_t *off, uint32_t h, uint32_t w)
{ return sao_apply_leaf<uint16_t>(4, r, st, nullptr, nullptr, band, off, h, w); }
extern "C" int svt_amd_SAOApplyEO_0_16bit(uint16_t *r, uint32_t st, uint16_t *left, int8_t *off, uint32_t h, uint32_t w)
{ return sao_apply_leaf<uint16_t>(0, r, st, left, nullptr, 0, off, h, w); }
extern "C" int svt_amd_SAOApplyEO_90_16bit(uint16_t *r, uint32_t st, uint16_t *upper, int8_t *off, uint32_t h, uint32_t w)
{ return sao_apply_leaf<uint16_t>(1, r, st, nullptr, upper, 0, off, h, w); }
extern "C" int svt_amd_SAOApplyEO_135_16bit(uint16_t *r, uint32_t st, uint16_t *left, uint16_t *upper, int8_t *off, uint32_t h, uint32_t w)
{ return sao_apply_leaf<uint16_t>(2, r, st, left, upper, 0, off, h, w); }
extern "C" int svt_amd_SAOApplyEO_45_16bit(uint16_t *r, uint32_t st, uint16_t *left, uint16_t *upper, int8_t *off, uint32_t h, uint32_t w)
{ return sao_apply_leaf<uint16_t>(3, r, st, left, upper, 0, off, h, w); }

/* pack / unpack leaves */
extern "C" void svt_amd_EB_ENC_msbPack2D(uint8_t *in8BitBuffer, uint32_t in8Stride, uint8_t *innBitBuffer,
                                         uint16_t *out16BitBuffer, uint32_t innStride, uint32_t outStride,
                                         uint32_t width, uint32_t height)
{
    DBuf a(in8BitBuffer, span(in8Stride, width, height)), b(innBitBuffer, span(innStride, width, height)),
        o(out16BitBuffer, span(outStride, width, height) * 2);
    if (!(a.ok && b.ok && o.ok))
        return;
    hipLaunchKernelGGL(k_pack, grid1d(width * height), dim3(256), 0, 0, a.d, in8Stride, b.d, innStride, (uint16_t *)o.d, outStride, width, height, 0);
    if (finish("EB_ENC_msbPack2D"))
        o.download(out16BitBuffer, span(outStride, width, height) * 2);
}
extern "C" void svt_amd_CompressedPackmsb(uint8_t *in8BitBuffer, uint32_t in8Stride, uint8_t *innBitBuffer,
                                          uint16_t *out16BitBuffer, uint32_t innStride, uint32_t outStride,
                                          uint32_t width, uint32_t height)
{
    const uint32_t w4 = width & ~3u; /* the reference packs width/4 groups */
    DBuf a(in8BitBuffer, span(in8Stride, width, height)), b(innBitBuffer, span(innStride, (width + 3) / 4, height)),
        o(out16BitBuffer, span(outStride, width, height) * 2);
    if (!(a.ok && b.ok && o.ok) || !w4)
        return;
    hipLaunchKernelGGL(k_pack, grid1d(w4 * height), dim3(256), 0, 0, a.d, in8Stride, b.d, innStride, (uint16_t *)o.d, outStride, w4, height, 1);
    if (finish("CompressedPackmsb"))
        o.download(out16BitBuffer, span(outStride, width, height) * 2);
}
extern "C" void svt_amd_CPack_C(const uint8_t *innBitBuffer, uint32_t innStride, uint8_t *inCompnBitBuffer,
                                uint32_t outStride, uint8_t *localCache, uint32_t width, uint32_t height)
{
    (void)localCache;
    DBuf a(innBitBuffer, span(innStride, width, height)), o(inCompnBitBuffer, span(outStride, width / 4, height));
    if (!(a.ok && o.ok))
        return;
    hipLaunchKernelGGL(k_cpack, grid1d(width / 4 * height), dim3(256), 0, 0, a.d, innStride, o.d, outStride, width, height);
    if (finish("CPack_C"))
        o.download(inCompnBitBuffer, span(outStride, width / 4, height));
}
static void unpack_leaf(uint16_t *in16, uint32_t inStride, uint8_t *out8, uint32_t out8Stride, uint8_t *outn,
                        uint32_t outnStride, uint32_t w, uint32_t h)
{
    DBuf a(in16, span(inStride, w, h) * 2), o8(out8, span(out8Stride, w, h)), on(outn ? outn : out8, outn ? span(outnStride, w, h) : 4);
    if (!(a.ok && o8.ok && on.ok))
        return;
    hipLaunchKernelGGL(k_unpack, grid1d(w * h), dim3(256), 0, 0, (const uint16_t *)a.d, inStride, o8.d, out8Stride, outn ? on.d : nullptr,
                       outnStride, w, h);
    if (!finish("unpack"))
        return;
    o8.download(out8, span(out8Stride, w, h));
    if (outn)
        on.download(outn, span(outnStride, w, h));
}
extern "C" void svt_amd_EB_ENC_msbUnPack2D(uint16_t *in16BitBuffer, uint32_t inStride, uint8_t *out8BitBuffer,
                                           uint8_t *outnBitBuffer, uint32_t out8Stride, uint32_t outnStride,
                                           uint32_t width, uint32_t height)
{
    unpack_leaf(in16BitBuffer, inStride, out8BitBuffer, out8Stride, outnBitBuffer, outnStride, width, height);
}
extern "C" void svt_amd_UnPack8BitData(uint16_t *in16BitBuffer, uint32_t inStride, uint8_t *out8BitBuffer,
                                       uint32_t out8Stride, uint32_t width, uint32_t height)
{
    unpack_leaf(in16BitBuffer, inStride, out8BitBuffer, out8Stride, nullptr, 0, width, height);
}
extern "C" void svt_amd_UnpackAvg(uint16_t *ref16L0, uint32_t refL0Stride, uint16_t *ref16L1, uint32_t refL1Stride,
                                  uint8_t *dstPtr, uint32_t dstStride, uint32_t width, uint32_t height)
{
    DBuf a(ref16L0, span(refL0Stride, width, height) * 2), b(ref16L1, span(refL1Stride, width, height) * 2),
        o(dstPtr, span(dstStride, width, height));
    if (!(a.ok && b.ok && o.ok))
        return;
    hipLaunchKernelGGL(k_unpack_avg, grid1d(width * height), dim3(256), 0, 0, (const uint16_t *)a.d, refL0Stride, (const uint16_t *)b.d,
                       refL1Stride, o.d, dstStride, width, height);
    if (finish("UnpackAvg"))
        o.download(dstPtr, span(dstStride, width, height));
}
