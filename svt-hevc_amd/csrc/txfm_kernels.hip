/*
 * Residual / transform / quantisation / distortion / SATD kernels of the EncDec half
 * (SURVEY.md 8a "EncDec" rows), batched over contiguous blocks in HBM:
 * block b of a size x size batch starts at base + b*size*size (row stride = size).
 *
 * Forward DCT/DST  - Transform32x32/16x16(+Estimate)/8x8/4x4, DstTransform4x4
 *                    (C_DEFAULT/EbTransforms_C.c:1602-1908; butterflies :269-1073)
 * Inverse          - InvTransform32x32/16x16/8x8/4x4, InvDstTransform4x4 (:1910-2119)
 * Quantisation     - QuantizeInvQuantize (:89-138), UpdateQiQCoef (:209-260)
 * Distortion       - FullDistortionKernel_32bit / CbfZero / Intra (C_DEFAULT/EbPictureOperators_C.c:385-480)
 * SATD             - Compute8x8Satd(_U8) (:481-642), Compute4x4Satd(_U8) (Codec/EbHmCode.c:41-215)
 * Picture ops      - ResidualKernel, PictureAdditionKernel, ZeroOutCoeffKernel, PictureCopyKernel (:83-360)
 *
 * The forward butterfly keeps the reference's exact integer widths: the low-precision
 * "Estimate" variants wrap the first one (16-point) / two (32-point) even/odd levels to
 * 16 bits, every pass truncates its output to int16.  Levels are evaluated in place in LDS
 * (one barrier per level), then every output is a short dot product with the HEVC matrix
 * row held in LDS.  Bound: HBM (4 bytes moved per coefficient, ~40 integer ops).
 */
#include "svt_amd_internal.h"
#include "txfm_device.h"
#include <cstring>

/* Forward DCT, register-resident: lane = one row of one N x N block, 64 / N blocks per wave, 4 waves per workgroup.
 * HBM: every residual row is one contiguous N*2-byte run per lane; results leave as one 2-byte store per lane and
 * output row (a wave covers 128 contiguous bytes).  LDS only carries the transpose between the two passes. */
template <int N>
__global__ __launch_bounds__(TX_THREADS) void k_fwd_dct(const int16_t *__restrict__ src, int16_t *__restrict__ dst,
                                                       uint32_t nblocks, int shift1, int shift2, int wrap_levels,
                                                       const uint8_t *__restrict__ only = nullptr)
{
    constexpr int UPW = 64 / N, UPB = UPW * (TX_THREADS / 64); /* units per wave / per workgroup */
    __shared__ int16_t tiles[UPB * TxRegTile<N>::UNIT];
    const int t = threadIdx.x, u = t / N, r = t - u * N;
    uint32_t b = blockIdx.x * UPB + u;
    if (only && b < nblocks && !only[b])
        b = nblocks; /* fix-up pass of the MFMA path: only the flagged blocks are transformed (and stored) */
    int x[N];
    if (b < nblocks) {
        const int16_t *row = src + (size_t)b * N * N + r * N;
        if (N >= 8) {
#pragma unroll
            for (int j = 0; j < N; j += 8) {
                const uint4 v = *(const uint4 *)(row + j);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++)
                    x[j + 2 * k] = (int16_t)(w[k] & 0xffffu), x[j + 2 * k + 1] = (int16_t)(w[k] >> 16);
            }
        } else {
            const uint2 v = *(const uint2 *)row;
            x[0] = (int16_t)(v.x & 0xffffu), x[1] = (int16_t)(v.x >> 16), x[2] = (int16_t)(v.y & 0xffffu), x[3] = (int16_t)(v.y >> 16);
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; j++)
            x[j] = 0;
    }
    fwd_2d_regs<N>(x, tiles + u * TxRegTile<N>::UNIT, r, shift1, shift2, wrap_levels);
    if (b < nblocks) {
        int16_t *out = dst + (size_t)b * N * N + r;
#pragma unroll
        for (int j = 0; j < N; j++)
            out[j * N] = (int16_t)x[j];
    }
}

/* 4x4 DST (Dst4 / DstInverse4): one thread per row, 64 blocks per workgroup */
__device__ __forceinline__ void dst4_fwd_row(const int16_t *x, int16_t *out, int ostride, int r, int shift)
{
    const int offset = (int16_t)(1 << (shift - 1));
    const int e0 = x[0] + x[3], o0 = x[1] + x[3], e1 = x[0] - x[1], o1 = 74 * x[2];
    out[0 * ostride + r] = (int16_t)((29 * e0 + 55 * o0 + o1 + offset) >> shift);
    out[2 * ostride + r] = (int16_t)((29 * e1 + 55 * e0 - o1 + offset) >> shift);
    out[1 * ostride + r] = (int16_t)(((74 * (x[0] + x[1] - x[3])) + offset) >> shift);
    out[3 * ostride + r] = (int16_t)((55 * e1 - 29 * o0 + o1 + offset) >> shift);
}
__device__ __forceinline__ void dst4_inv_col(const int16_t *in, int istride, int r, int16_t *out, int shift)
{
    const int offset = (int16_t)(1 << (shift - 1));
    const int c0 = in[r], c1 = in[istride + r], c2 = in[2 * istride + r], c3 = in[3 * istride + r];
    const int o0 = c0 + c2, o1 = c0 - c3, e0 = c2 + c3, e1 = 74 * c1;
    out[0] = (int16_t)clip16i((29 * o0 + 55 * e0 + e1 + offset) >> shift);
    out[1] = (int16_t)clip16i((55 * o1 - 29 * e0 + e1 + offset) >> shift);
    out[2] = (int16_t)clip16i(((74 * (c0 - c2 + c3)) + offset) >> shift);
    out[3] = (int16_t)clip16i((55 * o0 + 29 * o1 - e1 + offset) >> shift);
}
__device__ __forceinline__ void dst4_inv_col_regs(const int (&c)[4], int (&out)[4], int shift)
{
    const int offset = (int16_t)(1 << (shift - 1));
    const int o0 = c[0] + c[2], o1 = c[0] - c[3], e0 = c[2] + c[3], e1 = 74 * c[1];
    out[0] = clip16i((29 * o0 + 55 * e0 + e1 + offset) >> shift);
    out[1] = clip16i((55 * o1 - 29 * e0 + e1 + offset) >> shift);
    out[2] = clip16i(((74 * (c[0] - c[2] + c[3])) + offset) >> shift);
    out[3] = clip16i((55 * o0 + 29 * o1 - e1 + offset) >> shift);
}
template <int N>
__device__ __forceinline__ void dst4_inv_col_regs(const int (&)[N], int (&)[N], int) {} /* only the 4x4 unit has a DST */
__global__ __launch_bounds__(TX_THREADS) void k_dst4(const int16_t *__restrict__ src, int16_t *__restrict__ dst,
                                                    uint32_t nblocks, int shift1, int shift2, int inverse)
{
    __shared__ int16_t a[64][16], m[64][16];
    const int t = threadIdx.x, lb = t >> 2, r = t & 3;
    const uint32_t b = blockIdx.x * 64 + lb;
    if (b < nblocks)
        for (int j = 0; j < 4; j++)
            a[lb][r * 4 + j] = src[(size_t)b * 16 + r * 4 + j];
    __syncthreads();
    if (b < nblocks) {
        if (!inverse)
            dst4_fwd_row(&a[lb][r * 4], m[lb], 4, r, shift1);
        else
            dst4_inv_col(a[lb], 4, r, &m[lb][r * 4], shift1);
    }
    __syncthreads();
    if (b < nblocks) {
        if (!inverse)
            dst4_fwd_row(&m[lb][r * 4], a[lb], 4, r, shift2);
        else
            dst4_inv_col(m[lb], 4, r, &a[lb][r * 4], shift2);
    }
    __syncthreads();
    if (b < nblocks)
        for (int j = 0; j < 4; j++)
            dst[(size_t)b * 16 + r * 4 + j] = a[lb][r * 4 + j];
}

/* inverse DCT, register-resident: lane r = column r of the coefficient block in pass 1 (2-byte loads, a wave reads 128
 * contiguous bytes per coefficient row), row r of the result in pass 2 (one contiguous N*2-byte run per lane);
 * exact integer arithmetic per pass, clip to 16 bits. */
template <int N>
__global__ __launch_bounds__(TX_THREADS) void k_inv_dct(const int16_t *__restrict__ src, int16_t *__restrict__ dst,
                                                       uint32_t nblocks, int shift1, int shift2)
{
    constexpr int UPW = 64 / N, UPB = UPW * (TX_THREADS / 64), P = TxRegTile<N>::PITCH;
    __shared__ int16_t tiles[UPB * TxRegTile<N>::UNIT];
    const int t = threadIdx.x, u = t / N, r = t - u * N;
    const uint32_t b = blockIdx.x * UPB + u;
    int16_t *tile = tiles + u * TxRegTile<N>::UNIT;
    int c[N];
#pragma unroll
    for (int k = 0; k < N; k++)
        c[k] = b < nblocks ? (int)src[(size_t)b * N * N + k * N + r] : 0;
    /* pass 1: column r -> row r of the intermediate */
    inv_1d_regs<N>(c, shift1, [&](int j, int16_t v) { tile[r * P + j] = v; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < N; k++)
        c[k] = tile[k * P + r];
    int y[N];
    inv_1d_regs<N>(c, shift2, [&](int j, int16_t v) { y[j] = v; });
    if (b < nblocks) {
        int16_t *out = dst + (size_t)b * N * N + r * N;
        if (N >= 8) {
#pragma unroll
            for (int j = 0; j < N; j += 8) {
                uint4 v;
                v.x = (uint32_t)(uint16_t)y[j] | ((uint32_t)(uint16_t)y[j + 1] << 16);
                v.y = (uint32_t)(uint16_t)y[j + 2] | ((uint32_t)(uint16_t)y[j + 3] << 16);
                v.z = (uint32_t)(uint16_t)y[j + 4] | ((uint32_t)(uint16_t)y[j + 5] << 16);
                v.w = (uint32_t)(uint16_t)y[j + 6] | ((uint32_t)(uint16_t)y[j + 7] << 16);
                *(uint4 *)(out + j) = v;
            }
        } else {
            uint2 v;
            v.x = (uint32_t)(uint16_t)y[0] | ((uint32_t)(uint16_t)y[1] << 16);
            v.y = (uint32_t)(uint16_t)y[2] | ((uint32_t)(uint16_t)y[3] << 16);
            *(uint2 *)out = v;
        }
    }
}

/* Reconstruction of transform units, fused: EncodeGenerateRecon(16bit) (Codec/EbCodingLoop.c:1084, :1660) =
 * EncodeInvTransform (EbTransforms.c:3502; DC-only shortcut, inverse DCT, 4x4 inverse DST) + PictureAdditionKernel(16bit).
 * Lane r of a unit: column r of the coefficients in the first pass, row r of the residual after the second, then row r of
 * the prediction (one contiguous run) -> clipped sum -> one contiguous store.  The residual never leaves registers. */
struct ReconUnit { int32_t pred_off, recon_off; uint8_t only_dc, dst, pad[2]; }; /* = SvtAmdReconUnit */

template <int N, typename T>
__global__ __launch_bounds__(TX_THREADS) void k_recon_tu(const int16_t *__restrict__ coeff, const ReconUnit *__restrict__ units,
                                                        const T *__restrict__ pred, uint32_t predStride, T *__restrict__ recon,
                                                        uint32_t reconStride, uint32_t nunits, int shift1, int shift2)
{
    constexpr int UPW = 64 / N, UPB = UPW * (TX_THREADS / 64), P = TxRegTile<N>::PITCH;
    constexpr int maxv = sizeof(T) == 1 ? 255 : 1023;
    __shared__ int16_t tiles[UPB * TxRegTile<N>::UNIT];
    const int t = threadIdx.x, u = t / N, r = t - u * N;
    const uint32_t b = blockIdx.x * UPB + u;
    const bool live = b < nunits;
    int16_t *tile = tiles + u * TxRegTile<N>::UNIT;
    ReconUnit U = {0, 0, 0, 0, {0, 0}};
    if (live)
        U = units[b];
    int c[N], y[N];
#pragma unroll
    for (int k = 0; k < N; k++)
        c[k] = live ? (int)coeff[(size_t)b * N * N + k * N + r] : 0;
    if (N == 4 && U.dst) {
        /* InvDstTransform4x4: column r -> row r, twice (dst4_inv_col above) */
        dst4_inv_col_regs(c, y, shift1);
#pragma unroll
        for (int j = 0; j < 4; j++)
            tile[r * P + j] = (int16_t)y[j];
    } else {
        inv_1d_regs<N>(c, shift1, [&](int j, int16_t v) { tile[r * P + j] = v; });
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < N; k++)
        c[k] = tile[k * P + r];
    if (N == 4 && U.dst)
        dst4_inv_col_regs(c, y, shift2);
    else
        inv_1d_regs<N>(c, shift2, [&](int j, int16_t v) { y[j] = v; });
    if (U.only_dc) {
        /* EncodeInvTransform's shortcut: the twice scaled and clipped DC value everywhere */
        int v = (int)coeff[(size_t)b * N * N];
        v = clip16i((64 * v + (1 << (shift1 - 1))) >> shift1);
        v = clip16i((64 * (int16_t)v + (1 << (shift2 - 1))) >> shift2);
#pragma unroll
        for (int j = 0; j < N; j++)
            y[j] = v;
    }
    if (live) {
        int pr[N];
        load_row<N, T>(pred + U.pred_off + (size_t)r * predStride, pr);
#pragma unroll
        for (int j = 0; j < N; j++) {
            const int v = pr[j] + y[j];
            y[j] = v < 0 ? 0 : v > maxv ? maxv : v;
        }
        store_row<N, T>(recon + U.recon_off + (size_t)r * reconStride, y);
    }
}

/* ------------------------------------------------------------------------- */
/* quantisation, distortion, SATD, picture operators                          */
/* ------------------------------------------------------------------------- */

/* Wave-centric forms for power-of-two block sizes (16 .. 4096 coefficients, 16-byte aligned blocks): a lane moves
 * 8 coefficients per 16-byte access, a block is owned by n2/8 adjacent lanes (up to a whole wave, which then
 * iterates), per-block sums are segmented shuffles - no LDS, no barriers, several small blocks per wave. */
__device__ __forceinline__ uint32_t seg_sum(uint32_t v, int lanes)
{
    for (int o = 1; o < lanes; o <<= 1)
        v += __shfl_xor(v, o);
    return v;
}
__global__ __launch_bounds__(TX_THREADS) void k_quant_w(const int16_t *__restrict__ coeff, int16_t *__restrict__ q,
                                                       int16_t *__restrict__ rec, uint32_t *__restrict__ nz,
                                                       uint32_t nblocks, int n2, uint32_t qFunc, uint32_t q_offset,
                                                       int shiftedQBits, int shiftedFFunc, int iq_offset, int shiftNum)
{
    const int lane = threadIdx.x & 63, lpb = n2 >= 512 ? 64 : n2 >> 3, bpw = 64 / lpb, iters = n2 / (8 * lpb);
    const uint32_t gw = (blockIdx.x * TX_THREADS + threadIdx.x) >> 6, nw = (gridDim.x * TX_THREADS) >> 6;
    for (uint32_t wb = gw; wb * bpw < nblocks; wb += nw) {
        const uint32_t b = wb * bpw + lane / lpb;
        const int sub = lane % lpb;
        uint32_t mine = 0;
        if (b < nblocks) {
            for (int it = 0; it < iters; it++) {
                const size_t at = (size_t)b * n2 + (size_t)(it * lpb + sub) * 8;
                const uint4 cv = *(const uint4 *)(coeff + at);
                const uint32_t w[4] = {cv.x, cv.y, cv.z, cv.w};
                uint32_t qo[4], ro[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    uint32_t qq = 0, rr = 0;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int v = (int16_t)(w[k] >> (16 * h)), sign = v < 0 ? -1 : 1;
                        int tq = abs(v);
                        tq = (int)((uint32_t)tq * qFunc);
                        tq = (int)((uint32_t)tq + q_offset);
                        tq >>= shiftedQBits;
                        const int qv = clip16i(sign * tq);
                        mine += qv != 0;
                        qq |= (uint32_t)(uint16_t)qv << (16 * h);
                        rr |= (uint32_t)(uint16_t)clip16i(((qv * shiftedFFunc) + iq_offset) >> shiftNum) << (16 * h);
                    }
                    qo[k] = qq, ro[k] = rr;
                }
                *(uint4 *)(q + at) = make_uint4(qo[0], qo[1], qo[2], qo[3]);
                *(uint4 *)(rec + at) = make_uint4(ro[0], ro[1], ro[2], ro[3]);
            }
        }
        mine = seg_sum(mine, lpb);
        if (b < nblocks && sub == 0)
            nz[b] = mine;
    }
}
__global__ __launch_bounds__(TX_THREADS) void k_full_distortion_w(const int16_t *__restrict__ coeff,
                                                                 const int16_t *__restrict__ rec,
                                                                 unsigned long long *__restrict__ out, uint32_t nblocks,
                                                                 int n2, int mode)
{
    const int lane = threadIdx.x & 63, lpb = n2 >= 512 ? 64 : n2 >> 3, bpw = 64 / lpb, iters = n2 / (8 * lpb);
    const uint32_t gw = (blockIdx.x * TX_THREADS + threadIdx.x) >> 6, nw = (gridDim.x * TX_THREADS) >> 6;
    for (uint32_t wb = gw; wb * bpw < nblocks; wb += nw) {
        const uint32_t b = wb * bpw + lane / lpb;
        const int sub = lane % lpb;
        uint32_t res = 0, pred = 0;
        if (b < nblocks) {
            for (int it = 0; it < iters; it++) {
                const size_t at = (size_t)b * n2 + (size_t)(it * lpb + sub) * 8;
                const uint4 cv = *(const uint4 *)(coeff + at), rv = *(const uint4 *)(rec + at);
                const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w}, rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int16_t c = (int16_t)(cw[k] >> (16 * h)), d = (int16_t)(c - (int16_t)(rw[k] >> (16 * h)));
                        res += (uint32_t)(d * d);
                        pred += (uint32_t)(c * c);
                    }
            }
        }
        res = seg_sum(res, lpb), pred = seg_sum(pred, lpb);
        if (b < nblocks && sub == 0) {
            out[2 * b + 0] = mode == 1 ? pred : res;
            out[2 * b + 1] = mode == 2 ? res : pred;
        }
    }
}
static inline bool wave_form_ok(int n2, const void *a, const void *b, const void *c)
{
    return n2 >= 16 && n2 <= 4096 && (n2 & (n2 - 1)) == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}
static inline unsigned wave_form_grid(uint32_t n, int n2)
{
    const int lpb = n2 >= 512 ? 64 : n2 >> 3, bpw = 64 / lpb;
    const uint32_t waves = (n + bpw - 1) / bpw, wgs = (waves + 3) / 4;
    return wgs < 8192 ? wgs : 8192;
}

/* QuantizeInvQuantize over contiguous size x size blocks; nz[b] = non-zero count */
__global__ __launch_bounds__(TX_THREADS) void k_quant(const int16_t *__restrict__ coeff, int16_t *__restrict__ q,
                                                     int16_t *__restrict__ rec, uint32_t *__restrict__ nz,
                                                     uint32_t nblocks, int n2, uint32_t qFunc, uint32_t q_offset,
                                                     int shiftedQBits, int shiftedFFunc, int iq_offset, int shiftNum)
{
    __shared__ uint32_t cnt;
    for (uint32_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        if (threadIdx.x == 0)
            cnt = 0;
        __syncthreads();
        uint32_t mine = 0;
        for (int i = threadIdx.x; i < n2; i += TX_THREADS) {
            const int v = coeff[(size_t)b * n2 + i], sign = v < 0 ? -1 : 1;
            int tq = abs(v);
            tq = (int)((uint32_t)tq * qFunc);
            tq = (int)((uint32_t)tq + q_offset);
            tq >>= shiftedQBits;
            const int qv = clip16i(sign * tq);
            q[(size_t)b * n2 + i] = (int16_t)qv;
            mine += qv != 0;
            rec[(size_t)b * n2 + i] = (int16_t)clip16i(((qv * shiftedFFunc) + iq_offset) >> shiftNum);
        }
        atomicAdd(&cnt, mine);
        __syncthreads();
        if (threadIdx.x == 0)
            nz[b] = cnt;
        __syncthreads();
    }
}

/* FullDistortionKernel*_32bit: 32-bit modular sums of squared int16-truncated differences */
__global__ __launch_bounds__(TX_THREADS) void k_full_distortion(const int16_t *__restrict__ coeff,
                                                               const int16_t *__restrict__ rec,
                                                               unsigned long long *__restrict__ out, uint32_t nblocks,
                                                               int n2, int mode)
{
    __shared__ uint32_t acc[2];
    for (uint32_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        if (threadIdx.x < 2)
            acc[threadIdx.x] = 0;
        __syncthreads();
        uint32_t res = 0, pred = 0;
        for (int i = threadIdx.x; i < n2; i += TX_THREADS) {
            const int16_t c = coeff[(size_t)b * n2 + i], d = (int16_t)(c - rec[(size_t)b * n2 + i]);
            res += (uint32_t)(d * d);
            pred += (uint32_t)(c * c);
        }
        atomicAdd(&acc[0], res);
        atomicAdd(&acc[1], pred);
        __syncthreads();
        if (threadIdx.x == 0) {
            out[2 * b + 0] = mode == 1 ? acc[1] : acc[0];
            out[2 * b + 1] = mode == 2 ? acc[0] : acc[1];
        }
        __syncthreads();
    }
}

/* N x N Hadamard SATD, one block per group of N lanes: lane = row, row butterfly in registers,
 * column butterfly across lanes with xor shuffles; 16-bit arithmetic as in the reference */
template <int N>
__global__ __launch_bounds__(TX_THREADS) void k_satd(const int16_t *__restrict__ diff, const uint8_t *__restrict__ u8,
                                                    uint32_t u8stride, unsigned long long *__restrict__ satd,
                                                    long long *__restrict__ dc, uint32_t nblocks)
{
    const int lane = threadIdx.x & (N - 1);
    const uint32_t b = (blockIdx.x * TX_THREADS + threadIdx.x) / N;
    const bool ok = b < nblocks;
    int16_t v[N];
#pragma unroll
    for (int j = 0; j < N; j++)
        v[j] = !ok ? (int16_t)0 : diff ? diff[(size_t)b * N * N + lane * N + j]
                                       : (int16_t)u8[(size_t)b * N * N * 0 + (size_t)lane * u8stride + j + (size_t)b * N];
#pragma unroll
    for (int len = 1; len < N; len <<= 1)
#pragma unroll
        for (int i = 0; i < N; i += len << 1)
#pragma unroll
            for (int j = i; j < i + len; j++) {
                const int16_t s = (int16_t)(v[j] + v[j + len]), d = (int16_t)(v[j] - v[j + len]);
                v[j] = s, v[j + len] = d;
            }
#pragma unroll
    for (int len = 1; len < N; len <<= 1)
#pragma unroll
        for (int j = 0; j < N; j++) {
            const int o = __shfl_xor((int)v[j], len);
            v[j] = (lane & len) ? (int16_t)(o - v[j]) : (int16_t)(v[j] + o);
        }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < N; j++)
        s += (uint32_t)abs((int)v[j]);
    const int dc0 = v[0];
#pragma unroll
    for (int o = 1; o < N; o <<= 1)
        s += __shfl_xor(s, o);
    if (ok && lane == 0) {
        satd[b] = N == 8 ? ((unsigned long long)s + 2) >> 2 : ((unsigned long long)s + 1) >> 1;
        if (dc)
            dc[b] = dc0;
    }
}

__global__ void k_residual(const uint8_t *in, uint32_t is, const uint8_t *pred, uint32_t ps, int16_t *res,
                           uint32_t rs, uint32_t w, uint32_t h)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / w, x = i - y * w;
        res[(size_t)y * rs + x] = (int16_t)((int16_t)in[(size_t)y * is + x] - (int16_t)pred[(size_t)y * ps + x]);
    }
}
__global__ void k_addition(const uint8_t *pred, uint32_t ps, const int16_t *res, uint32_t rs, uint8_t *rec,
                           uint32_t cs, uint32_t w, uint32_t h)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / w, x = i - y * w;
        const int v = (int)res[(size_t)y * rs + x] + pred[(size_t)y * ps + x];
        rec[(size_t)y * cs + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

/* ------------------------------------------------------------------------- */
/* launchers (device pointers)                                                */
/* ------------------------------------------------------------------------- */

static int fwd_shifts(int kind, int size, uint32_t inc, int *s1, int *s2, int *wrap)
{
    const int lg = size == 32 ? 5 : size == 16 ? 4 : size == 8 ? 3 : size == 4 ? 2 : -1;
    if (lg < 0 || kind < 0 || kind > 2 || (kind == 2 && size != 4) || inc > 4)
        return SVT_AMD_ERR_BAD_PARAM;
    const int est = kind == 1 && size >= 16;
    *s1 = est ? (size == 32 ? 6 : 4) + (int)inc : lg - 1 + (int)inc;
    *s2 = est ? 9 : lg + 6;
    *wrap = est ? (size == 32 ? 2 : 1) : 0;
    return SVT_AMD_OK;
}

int svt_amd_launch_fwd_transform(hipStream_t st, int kind, int size, uint32_t inc, const int16_t *d_res,
                                 int16_t *d_coeff, uint32_t n)
{
    int s1, s2, wrap;
    int rc = fwd_shifts(kind, size, inc, &s1, &s2, &wrap);
    if (rc || !n)
        return rc ? rc : SVT_AMD_ERR_BAD_PARAM;
    if (kind == 2)
        hipLaunchKernelGGL(k_dst4, dim3((n + 63) / 64), dim3(TX_THREADS), 0, st, d_res, d_coeff, n, s1, s2, 0);
    else if (size == 32)
        hipLaunchKernelGGL(k_fwd_dct<32>, dim3((n + 7) / 8), dim3(TX_THREADS), 0, st, d_res, d_coeff, n, s1, s2, wrap, (const uint8_t *)nullptr);
    else if (size == 16)
        hipLaunchKernelGGL(k_fwd_dct<16>, dim3((n + 15) / 16), dim3(TX_THREADS), 0, st, d_res, d_coeff, n, s1, s2, wrap, (const uint8_t *)nullptr);
    else if (size == 8)
        hipLaunchKernelGGL(k_fwd_dct<8>, dim3((n + 31) / 32), dim3(TX_THREADS), 0, st, d_res, d_coeff, n, s1, s2, wrap, (const uint8_t *)nullptr);
    else
        hipLaunchKernelGGL(k_fwd_dct<4>, dim3((n + 63) / 64), dim3(TX_THREADS), 0, st, d_res, d_coeff, n, s1, s2, wrap, (const uint8_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

int svt_amd_launch_fwd_transform_flagged(hipStream_t st, int kind, int size, uint32_t inc, const int16_t *d_res, int16_t *d_coeff, uint32_t n,
                                         const uint8_t *d_only)
{
    int s1, s2, wrap;
    int rc = fwd_shifts(kind, size, inc, &s1, &s2, &wrap);
    if (rc || !n || size != 32)
        return rc ? rc : SVT_AMD_ERR_BAD_PARAM;
    hipLaunchKernelGGL(k_fwd_dct<32>, dim3((n + 7) / 8), dim3(TX_THREADS), 0, st, d_res, d_coeff, n, s1, s2, wrap, d_only);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

int svt_amd_launch_inv_transform(hipStream_t st, int kind, int size, uint32_t inc, const int16_t *d_coeff,
                                 int16_t *d_res, uint32_t n)
{
    if (!n || inc > 4 || !(size == 4 || size == 8 || size == 16 || size == 32) || (kind == 2 && size != 4) ||
        (kind != 0 && kind != 2))
        return SVT_AMD_ERR_BAD_PARAM;
    const int s1 = 7, s2 = 12 - (int)inc; /* SHIFT_INV_1ST, SHIFT_INV_2ND - bitIncrement */
    if (kind == 2)
        hipLaunchKernelGGL(k_dst4, dim3((n + 63) / 64), dim3(TX_THREADS), 0, st, d_coeff, d_res, n, s1, s2, 1);
    else if (size == 32)
        hipLaunchKernelGGL(k_inv_dct<32>, dim3((n + 7) / 8), dim3(TX_THREADS), 0, st, d_coeff, d_res, n, s1, s2);
    else if (size == 16)
        hipLaunchKernelGGL(k_inv_dct<16>, dim3((n + 15) / 16), dim3(TX_THREADS), 0, st, d_coeff, d_res, n, s1, s2);
    else if (size == 8)
        hipLaunchKernelGGL(k_inv_dct<8>, dim3((n + 31) / 32), dim3(TX_THREADS), 0, st, d_coeff, d_res, n, s1, s2);
    else
        hipLaunchKernelGGL(k_inv_dct<4>, dim3((n + 63) / 64), dim3(TX_THREADS), 0, st, d_coeff, d_res, n, s1, s2);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

int svt_amd_launch_quant(hipStream_t st, int size, uint32_t qFunc, uint32_t q_offset, int shiftedQBits,
                         int shiftedFFunc, int iq_offset, int shiftNum, const int16_t *d_coeff, int16_t *d_q,
                         int16_t *d_rec, uint32_t *d_nz, uint32_t n)
{
    if (!n || size < 4 || size > 64)
        return SVT_AMD_ERR_BAD_PARAM;
    if (wave_form_ok(size * size, d_coeff, d_q, d_rec))
        hipLaunchKernelGGL(k_quant_w, dim3(wave_form_grid(n, size * size)), dim3(TX_THREADS), 0, st, d_coeff, d_q, d_rec, d_nz,
                           n, size * size, qFunc, q_offset, shiftedQBits, shiftedFFunc, iq_offset, shiftNum);
    else
        hipLaunchKernelGGL(k_quant, dim3(n < 4096 ? n : 4096), dim3(TX_THREADS), 0, st, d_coeff, d_q, d_rec, d_nz, n,
                           size * size, qFunc, q_offset, shiftedQBits, shiftedFFunc, iq_offset, shiftNum);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

int svt_amd_launch_full_distortion(hipStream_t st, int size, int mode, const int16_t *d_coeff, const int16_t *d_rec,
                                   unsigned long long *d_out, uint32_t n)
{
    if (!n || mode < 0 || mode > 2)
        return SVT_AMD_ERR_BAD_PARAM;
    if (wave_form_ok(size * size, d_coeff, d_rec, d_rec))
        hipLaunchKernelGGL(k_full_distortion_w, dim3(wave_form_grid(n, size * size)), dim3(TX_THREADS), 0, st, d_coeff, d_rec,
                           d_out, n, size * size, mode);
    else
        hipLaunchKernelGGL(k_full_distortion, dim3(n < 4096 ? n : 4096), dim3(TX_THREADS), 0, st, d_coeff, d_rec, d_out, n,
                           size * size, mode);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* element-count form used by the leaf wrapper (rectangular areas) */
int svt_amd_launch_full_distortion_n(hipStream_t st, int n2, int mode, const int16_t *d_coeff, const int16_t *d_rec,
                                     unsigned long long *d_out, uint32_t n)
{
    if (!n || mode < 0 || mode > 2 || n2 < 1)
        return SVT_AMD_ERR_BAD_PARAM;
    hipLaunchKernelGGL(k_full_distortion, dim3(n < 4096 ? n : 4096), dim3(TX_THREADS), 0, st, d_coeff, d_rec, d_out, n, n2, mode);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

int svt_amd_launch_satd(hipStream_t st, int size, const int16_t *d_diff, const uint8_t *d_u8, uint32_t u8stride,
                        unsigned long long *d_satd, long long *d_dc, uint32_t n)
{
    if (!n || (size != 4 && size != 8))
        return SVT_AMD_ERR_BAD_PARAM;
    const uint32_t per = TX_THREADS / (uint32_t)size;
    if (size == 8)
        hipLaunchKernelGGL(k_satd<8>, dim3((n + per - 1) / per), dim3(TX_THREADS), 0, st, d_diff, d_u8, u8stride, d_satd, d_dc, n);
    else
        hipLaunchKernelGGL(k_satd<4>, dim3((n + per - 1) / per), dim3(TX_THREADS), 0, st, d_diff, d_u8, u8stride, d_satd, d_dc, n);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

int svt_amd_launch_residual(hipStream_t st, const uint8_t *in, uint32_t is, const uint8_t *pred, uint32_t ps,
                            int16_t *res, uint32_t rs, uint32_t w, uint32_t h)
{
    hipLaunchKernelGGL(k_residual, dim3((w * h + 255) / 256 < 2048 ? (w * h + 255) / 256 : 2048), dim3(256), 0, st, in, is,
                       pred, ps, res, rs, w, h);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
int svt_amd_launch_addition(hipStream_t st, const uint8_t *pred, uint32_t ps, const int16_t *res, uint32_t rs,
                            uint8_t *rec, uint32_t cs, uint32_t w, uint32_t h)
{
    hipLaunchKernelGGL(k_addition, dim3((w * h + 255) / 256 < 2048 ? (w * h + 255) / 256 : 2048), dim3(256), 0, st, pred,
                       ps, res, rs, rec, cs, w, h);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* ---- batched C-ABI (device pointers, context stream) ---- */
extern "C" int svt_amd_fwd_transform_batch(SvtAmdContext *ctx, int kind, int size, uint32_t bitIncrement,
                                           const int16_t *d_residual, int16_t *d_coeff, uint32_t nblocks)
{
    if (!ctx || !d_residual || !d_coeff)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    return svt_amd_launch_fwd_transform(ctx->stream, kind, size, bitIncrement, d_residual, d_coeff, nblocks);
}
extern "C" int svt_amd_inv_transform_batch(SvtAmdContext *ctx, int kind, int size, uint32_t bitIncrement,
                                           const int16_t *d_coeff, int16_t *d_residual, uint32_t nblocks)
{
    if (!ctx || !d_residual || !d_coeff)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    return svt_amd_launch_inv_transform(ctx->stream, kind, size, bitIncrement, d_coeff, d_residual, nblocks);
}
extern "C" int svt_amd_quantize_batch(SvtAmdContext *ctx, int size, uint32_t qFunc, uint32_t q_offset,
                                      int32_t shiftedQBits, int32_t shiftedFFunc, int32_t iq_offset, int32_t shiftNum,
                                      const int16_t *d_coeff, int16_t *d_quant, int16_t *d_recon, uint32_t *d_nz,
                                      uint32_t nblocks)
{
    if (!ctx || !d_coeff || !d_quant || !d_recon || !d_nz)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    return svt_amd_launch_quant(ctx->stream, size, qFunc, q_offset, shiftedQBits, shiftedFFunc, iq_offset, shiftNum,
                                d_coeff, d_quant, d_recon, d_nz, nblocks);
}
extern "C" int svt_amd_full_distortion_batch(SvtAmdContext *ctx, int size, int mode, const int16_t *d_coeff,
                                             const int16_t *d_recon, uint64_t *d_result, uint32_t nblocks)
{
    if (!ctx || !d_coeff || !d_recon || !d_result)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    return svt_amd_launch_full_distortion(ctx->stream, size, mode, d_coeff, d_recon, (unsigned long long *)d_result, nblocks);
}
extern "C" int svt_amd_satd_batch(SvtAmdContext *ctx, int size, const int16_t *d_diff, uint64_t *d_satd,
                                  uint32_t nblocks)
{
    if (!ctx || !d_diff || !d_satd)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    return svt_amd_launch_satd(ctx->stream, size, d_diff, nullptr, 0, (unsigned long long *)d_satd, nullptr, nblocks);
}

extern "C" int svt_amd_recon_tu_batch(SvtAmdContext *ctx, int bytes_per_sample, int size, const int16_t *d_coeff,
                                      const SvtAmdReconUnit *d_units, const void *d_pred, uint32_t predStride, void *d_recon,
                                      uint32_t reconStride, uint32_t nunits)
{
    if (!ctx || !d_coeff || !d_units || !d_pred || !d_recon || !nunits || (bytes_per_sample != 1 && bytes_per_sample != 2) ||
        !(size == 4 || size == 8 || size == 16 || size == 32))
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    const int s1 = 7, s2 = bytes_per_sample == 1 ? 12 : 10; /* SHIFT_INV_1ST, SHIFT_INV_2ND - bitIncrement (0 / 2) */
    const ReconUnit *un = (const ReconUnit *)d_units;
#define SVT_RECON_LAUNCH(N, T)                                                                                               \
    hipLaunchKernelGGL((k_recon_tu<N, T>), dim3((nunits + (64 / N) * 4 - 1) / ((64 / N) * 4)), dim3(TX_THREADS), 0, ctx->stream, \
                       d_coeff, un, (const T *)d_pred, predStride, (T *)d_recon, reconStride, nunits, s1, s2)
    if (bytes_per_sample == 1) {
        if (size == 32) SVT_RECON_LAUNCH(32, uint8_t);
        else if (size == 16) SVT_RECON_LAUNCH(16, uint8_t);
        else if (size == 8) SVT_RECON_LAUNCH(8, uint8_t);
        else SVT_RECON_LAUNCH(4, uint8_t);
    } else {
        if (size == 32) SVT_RECON_LAUNCH(32, uint16_t);
        else if (size == 16) SVT_RECON_LAUNCH(16, uint16_t);
        else if (size == 8) SVT_RECON_LAUNCH(8, uint16_t);
        else SVT_RECON_LAUNCH(4, uint16_t);
    }
#undef SVT_RECON_LAUNCH
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* Host-pointer form for one unit (the per-call binding of integration/svt_hook_me.c): operands are staged through a
 * scratch buffer of the context's device; blocking; callers serialise per context. */
extern "C" int svt_amd_recon_tu(SvtAmdContext *ctx, int bytes_per_sample, int size, int only_dc, int dst, const int16_t *coeff,
                                uint32_t coeffStride, const void *pred, uint32_t predStride, void *recon, uint32_t reconStride)
{
    if (!ctx || !coeff || !pred || !recon || (bytes_per_sample != 1 && bytes_per_sample != 2) ||
        !(size == 4 || size == 8 || size == 16 || size == 32) || coeffStride < (uint32_t)size || predStride < (uint32_t)size ||
        reconStride < (uint32_t)size)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *d_scratch = nullptr; /* unit | coeff | pred -> recon (in place) */
    const size_t o_unit = 0, o_coeff = 64, o_pix = o_coeff + 2048, total = o_pix + 2048;
    {
        int rc_s = svt_amd_ctx_scratch(ctx, total, &d_scratch);
        if (rc_s)
            return rc_s;
    }
    const size_t bps = (size_t)bytes_per_sample;
    int16_t hc[32 * 32];
    uint8_t hp[32 * 32 * 2];
    for (int y = 0; y < size; y++) {
        ::memcpy(hc + y * size, coeff + (size_t)y * coeffStride, (size_t)size * 2);
        ::memcpy(hp + (size_t)y * size * bps, (const uint8_t *)pred + (size_t)y * predStride * bps, (size_t)size * bps);
    }
    SvtAmdReconUnit u = {0, 0, (uint8_t)(only_dc != 0), (uint8_t)(dst != 0), {0, 0}};
    HIP_TRY(hipMemcpyAsync(d_scratch + o_unit, &u, sizeof(u), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_scratch + o_coeff, hc, (size_t)size * size * 2, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_scratch + o_pix, hp, (size_t)size * size * bps, hipMemcpyHostToDevice, ctx->stream));
    int rc = svt_amd_recon_tu_batch(ctx, bytes_per_sample, size, (const int16_t *)(d_scratch + o_coeff),
                                    (const SvtAmdReconUnit *)(d_scratch + o_unit), d_scratch + o_pix, (uint32_t)size,
                                    d_scratch + o_pix, (uint32_t)size, 1);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(hp, d_scratch + o_pix, (size_t)size * size * bps, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int y = 0; y < size; y++)
        ::memcpy((uint8_t *)recon + (size_t)y * reconStride * bps, hp + (size_t)y * size * bps, (size_t)size * bps);
    return SVT_AMD_OK;
}

/* ------------------------------------------------------------------------- */
/* Quantiser of the final encode pass: UnifiedQuantizeInvQuantize (Codec/EbTransforms.c:2978) without RDOQ / masking */
/* ------------------------------------------------------------------------- */
struct QuantUnit { uint8_t size, qp, bit_depth, slice_type, shape, clean_sparse, enable_cb_flag, contouring_flag, component,
                   temporal_layer, pad[2]; uint32_t dz_offset; }; /* = SvtAmdQuantUnit */

/* one workgroup per unit; the quantised block is mirrored in LDS for the two neighbourhood-dependent post-passes
 * (isolated-coefficient clean-up per 4x4 block, UpdateQiQCoef) */
__global__ __launch_bounds__(256) void k_unified_quant(const QuantUnit *__restrict__ units, const int16_t *__restrict__ coeff,
                                                      int16_t *__restrict__ quant, int16_t *__restrict__ recon,
                                                      uint32_t *__restrict__ nzOut)
{
    __shared__ int16_t q[32 * 32], r[32 * 32];
    __shared__ unsigned s_nz;
    const QuantUnit U = units[blockIdx.x];
    const int t = threadIdx.x, N = U.size, lg = 31 - __clz(N);
    const size_t base = (size_t)blockIdx.x * 1024;
    const int qpRem = U.qp % 6, qpPer = U.qp / 6;
    const uint32_t QF = qpRem == 0 ? 26214u : qpRem == 1 ? 23302u : qpRem == 2 ? 20560u : qpRem == 3 ? 18396u : qpRem == 4 ? 16384u : 14564u;
    const int FFv = qpRem == 0 ? 40 : qpRem == 1 ? 45 : qpRem == 2 ? 51 : qpRem == 3 ? 57 : qpRem == 4 ? 64 : 72;
    const int tshift = 15 - U.bit_depth - lg, shiftedQBits = 14 + qpPer + tshift;
    const uint32_t q_offset = ((U.slice_type == 2 || U.slice_type == 3) ? 171u : 85u) << (shiftedQBits - 9);
    const int shiftedFFunc = qpPer > 8 ? FFv << (qpPer - 2) : FFv << qpPer;
    const int shiftNum = qpPer > 8 ? 20 - 14 - tshift - 2 : 20 - 14 - tshift, iq_offset = 1 << (shiftNum - 1);
    const bool only_dc = U.shape == 3;
    const uint32_t offs = (!only_dc && U.dz_offset) ? (uint32_t)(U.dz_offset * (1u << shiftedQBits) / 20) : q_offset;
    const int area = only_dc ? 1 : N >> U.shape;
    if (t == 0)
        s_nz = 0;
    __syncthreads();
    unsigned nz = 0;
    for (int i = t; i < area * area; i += 256) {
        const int y = i / area, x = i - y * area, v = coeff[base + y * N + x], sign = v < 0 ? -1 : 1;
        int tq = abs(v);
        tq = (int)((uint32_t)tq * QF);
        tq = (int)((uint32_t)tq + offs);
        tq >>= shiftedQBits;
        const int qv = clip16i(sign * tq);
        q[y * N + x] = (int16_t)qv;
        r[y * N + x] = (int16_t)clip16i(((qv * shiftedFFunc) + iq_offset) >> shiftNum);
        nz += qv != 0;
    }
    for (int o = 32; o > 0; o >>= 1)
        nz += __shfl_xor(nz, o);
    if ((t & 63) == 0)
        atomicAdd(&s_nz, nz);
    __syncthreads();
    if (!only_dc) {
        /* zero coefficients that are alone in their 4x4 block, except position (0,0) (:3190-3232) */
        if (U.clean_sparse && N >= 8 && s_nz && U.slice_type != 2 && area >= 4) {
            const int nb = area >> 2;
            unsigned removed = 0;
            for (int b = t; b < nb * nb; b += 256) {
                const int by = b / nb, bx = b - by * nb;
                int cnt = 0, loc = -1;
                for (int y = 0; y < 4; y++)
                    for (int x = 0; x < 4; x++)
                        if (q[(4 * by + y) * N + 4 * bx + x])
                            cnt++, loc = (4 * by + y) * N + 4 * bx + x;
                if (cnt == 1 && loc != 0)
                    q[loc] = 0, r[loc] = 0, removed++;
            }
            if (removed)
                atomicSub(&s_nz, removed);
            __syncthreads();
        }
        if (t == 0) { /* UpdateQiQCoef (C_DEFAULT/EbTransforms_C.c:209-260) */
            unsigned n = s_nz;
            if (n < 10 && U.contouring_flag && U.slice_type == 2 && U.temporal_layer == 0 && U.component == 0) {
                const int loc = (area - 1) + (area - 1) * N;
                if (q[loc] == 0)
                    n++, q[loc] = 1, r[loc] = (int16_t)((int16_t)((1 * shiftedFFunc) + iq_offset) >> shiftNum);
            }
            if (n == 0 && U.enable_cb_flag == 1) {
                const int loc = (area - 2) * N + (area - 1);
                n = 1, q[loc] = 1, r[loc] = (int16_t)((int16_t)((1 * shiftedFFunc) + iq_offset) >> shiftNum);
            }
            s_nz = n;
        }
        __syncthreads();
    }
    for (int i = t; i < area * area; i += 256) {
        const int y = i / area, x = i - y * area;
        quant[base + y * N + x] = q[y * N + x];
        recon[base + y * N + x] = r[y * N + x];
    }
    if (t == 0)
        nzOut[blockIdx.x] = s_nz;
}

extern "C" int svt_amd_unified_quantize_batch(SvtAmdContext *ctx, const SvtAmdQuantUnit *d_units, const int16_t *d_coeff,
                                              int16_t *d_quant, int16_t *d_recon, uint32_t *d_nz, uint32_t nunits)
{
    static_assert(sizeof(QuantUnit) == sizeof(SvtAmdQuantUnit), "unit layout");
    if (!ctx || !d_units || !d_coeff || !d_quant || !d_recon || !d_nz || !nunits)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_unified_quant, dim3(nunits), dim3(256), 0, ctx->stream, (const QuantUnit *)d_units, d_coeff, d_quant, d_recon, d_nz);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

extern "C" int svt_amd_unified_quantize(SvtAmdContext *ctx, const SvtAmdQuantUnit *unit, const int16_t *coeff, uint32_t coeffStride,
                                        int16_t *quant, int16_t *recon, uint32_t *nz)
{
    if (!ctx || !unit || !coeff || !quant || !recon || !nz || !(unit->size == 4 || unit->size == 8 || unit->size == 16 || unit->size == 32) ||
        coeffStride < unit->size || unit->shape > 3 || (unit->bit_depth != 8 && unit->bit_depth != 10) || unit->qp > 51 ||
        (unit->shape != 3 && (unit->size >> unit->shape) < 2))
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *d_scratch = nullptr; /* unit | nz | coeff | quant | recon ; callers serialise per context */
    const size_t o_unit = 0, o_nz = 64, o_c = 128, o_q = o_c + 2048, o_r = o_q + 2048, total = o_r + 2048;
    {
        int rc_s = svt_amd_ctx_scratch(ctx, total, &d_scratch);
        if (rc_s)
            return rc_s;
    }
    const int N = unit->size, area = unit->shape == 3 ? 1 : N >> unit->shape;
    int16_t hc[32 * 32], hq[32 * 32], hr[32 * 32];
    for (int y = 0; y < N; y++)
        ::memcpy(hc + y * N, coeff + (size_t)y * coeffStride, (size_t)N * 2);
    HIP_TRY(hipMemcpyAsync(d_scratch + o_unit, unit, sizeof(*unit), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_scratch + o_c, hc, (size_t)N * N * 2, hipMemcpyHostToDevice, ctx->stream));
    int rc = svt_amd_unified_quantize_batch(ctx, (const SvtAmdQuantUnit *)(d_scratch + o_unit), (const int16_t *)(d_scratch + o_c),
                                            (int16_t *)(d_scratch + o_q), (int16_t *)(d_scratch + o_r), (uint32_t *)(d_scratch + o_nz), 1);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(nz, d_scratch + o_nz, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hq, d_scratch + o_q, (size_t)N * N * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hr, d_scratch + o_r, (size_t)N * N * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int y = 0; y < area; y++) { /* like the reference, only the active area is written */
        ::memcpy(quant + (size_t)y * coeffStride, hq + y * N, (size_t)area * 2);
        ::memcpy(recon + (size_t)y * coeffStride, hr + y * N, (size_t)area * 2);
    }
    return SVT_AMD_OK;
}

/* ------------------------------------------------------------------------- */
/* The whole transform unit of the final encode pass in one kernel             */
/* ------------------------------------------------------------------------- */
/* EncodeLoop (Codec/EbCodingLoop.c:651: PictureResidual -> EstimateTransform -> UnifiedQuantizeInvQuantize) followed by
 * EncodeGenerateRecon (:1084: EncodeInvTransform -> PictureAdditionKernel) for one plane of one unit, default coefficient
 * shape, no RDOQ / masking, DCT units (the 4x4 luma DST unit goes through the separate kernels).  Lane r of a unit: row r of
 * source and prediction -> residual -> forward transform in registers -> column r of the coefficients -> quantised and
 * inverse-quantised in registers (the quantised column is the one HBM output besides the picture) -> column r is exactly
 * what the inverse transform's first pass wants -> row r of the residual -> + prediction, clipped -> reconstruction row r.
 * HBM per sample: 1 B source + 1 B prediction in, 2 B coefficients + 1 B reconstruction out (8-bit). */
struct EncodeUnit { int32_t src_off, rec_off; uint8_t qp, slice_type, pad[2]; uint32_t dz_offset; }; /* = SvtAmdEncodeUnit */

template <int N, typename T>
__global__ __launch_bounds__(TX_THREADS) void k_encode_tu(const EncodeUnit *__restrict__ units, const T *__restrict__ src,
                                                         uint32_t srcStride, T *__restrict__ rec, uint32_t recStride,
                                                         int16_t *__restrict__ quantOut, uint32_t *__restrict__ nzOut,
                                                         uint32_t nunits, int fs1, int fs2, int wrap, int is1, int is2)
{
    constexpr int UPW = 64 / N, UPB = UPW * (TX_THREADS / 64), P = TxRegTile<N>::PITCH;
    constexpr int maxv = sizeof(T) == 1 ? 255 : 1023, LG = N == 32 ? 5 : N == 16 ? 4 : N == 8 ? 3 : 2;
    constexpr int depth = sizeof(T) == 1 ? 8 : 10;
    __shared__ int16_t tiles[UPB * TxRegTile<N>::UNIT];
    const int t = threadIdx.x, u = t / N, r = t - u * N;
    const uint32_t b = blockIdx.x * UPB + u;
    const bool live = b < nunits;
    int16_t *tile = tiles + u * TxRegTile<N>::UNIT;
    EncodeUnit U = {0, 0, 0, 0, {0, 0}, 0};
    if (live)
        U = units[b];
    int x[N], pred[N];
    if (live) {
        load_row<N, T>(rec + U.rec_off + (size_t)r * recStride, pred);
        load_row<N, T>(src + U.src_off + (size_t)r * srcStride, x);
#pragma unroll
        for (int j = 0; j < N; j++)
            x[j] -= pred[j];
    } else {
#pragma unroll
        for (int j = 0; j < N; j++)
            x[j] = 0, pred[j] = 0;
    }
    fwd_2d_regs<N>(x, tile, r, fs1, fs2, wrap); /* x[j] = coefficient (j, r) */
    /* UnifiedQuantizeInvQuantize, default shape (EbTransforms.c:3097-3160) */
    const int qpRem = U.qp % 6, qpPer = U.qp / 6;
    const uint32_t QF = qpRem == 0 ? 26214u : qpRem == 1 ? 23302u : qpRem == 2 ? 20560u : qpRem == 3 ? 18396u : qpRem == 4 ? 16384u : 14564u;
    const int FFv = qpRem == 0 ? 40 : qpRem == 1 ? 45 : qpRem == 2 ? 51 : qpRem == 3 ? 57 : qpRem == 4 ? 64 : 72;
    const int tshift = 15 - depth - LG, shiftedQBits = 14 + qpPer + tshift;
    const uint32_t q_offset = ((U.slice_type == 2 || U.slice_type == 3) ? 171u : 85u) << (shiftedQBits - 9);
    const uint32_t offs = U.dz_offset ? (uint32_t)(U.dz_offset * (1u << shiftedQBits) / 20) : q_offset;
    const int shiftedFFunc = qpPer > 8 ? FFv << (qpPer - 2) : FFv << qpPer;
    const int shiftNum = qpPer > 8 ? 20 - 14 - tshift - 2 : 20 - 14 - tshift, iq_offset = 1 << (shiftNum - 1);
    unsigned nz = 0;
    int c[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        const int v = x[j], sign = v < 0 ? -1 : 1;
        int tq = abs(v);
        tq = (int)((uint32_t)tq * QF);
        tq = (int)((uint32_t)tq + offs);
        tq >>= shiftedQBits;
        const int qv = clip16i(sign * tq);
        c[j] = clip16i(((qv * shiftedFFunc) + iq_offset) >> shiftNum);
        nz += qv != 0;
        if (live)
            quantOut[(size_t)b * N * N + j * N + r] = (int16_t)qv;
    }
#pragma unroll
    for (int o = 1; o < N; o <<= 1)
        nz += __shfl_xor(nz, o);
    if (live && r == 0)
        nzOut[b] = nz;
    /* EncodeInvTransform + PictureAdditionKernel; the tile is free again (both forward passes are through with it) */
    __builtin_amdgcn_wave_barrier();
    inv_1d_regs<N>(c, is1, [&](int j, int16_t v) { tile[r * P + j] = v; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < N; k++)
        c[k] = tile[k * P + r];
    int y[N];
    inv_1d_regs<N>(c, is2, [&](int j, int16_t v) { y[j] = v; });
    if (live) {
#pragma unroll
        for (int j = 0; j < N; j++) {
            const int v = pred[j] + y[j];
            y[j] = v < 0 ? 0 : v > maxv ? maxv : v;
        }
        store_row<N, T>(rec + U.rec_off + (size_t)r * recStride, y);
    }
}

extern "C" int svt_amd_encode_tu_batch(SvtAmdContext *ctx, int bytes_per_sample, int size, const SvtAmdEncodeUnit *d_units,
                                       const void *d_src, uint32_t srcStride, void *d_rec, uint32_t recStride, int16_t *d_quant,
                                       uint32_t *d_nz, uint32_t nunits)
{
    static_assert(sizeof(EncodeUnit) == sizeof(SvtAmdEncodeUnit), "unit layout");
    if (!ctx || !d_units || !d_src || !d_rec || !d_quant || !d_nz || !nunits || (bytes_per_sample != 1 && bytes_per_sample != 2) ||
        !(size == 4 || size == 8 || size == 16 || size == 32))
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    const int inc = bytes_per_sample == 1 ? 0 : 2;
    /* EstimateTransform (32: 6+inc / 9 wrap 2, 16: 4+inc / 9 wrap 1, 8: 2+inc / 9, 4: 1+inc / 8), EncodeInvTransform 7 / 12-inc */
    const int fs1 = (size == 32 ? 6 : size == 16 ? 4 : size == 8 ? 2 : 1) + inc, fs2 = size == 4 ? 8 : 9, wrap = size == 32 ? 2 : size == 16 ? 1 : 0;
    const EncodeUnit *un = (const EncodeUnit *)d_units;
#define SVT_ENC_LAUNCH(N, T)                                                                                                  \
    hipLaunchKernelGGL((k_encode_tu<N, T>), dim3((nunits + (64 / N) * 4 - 1) / ((64 / N) * 4)), dim3(TX_THREADS), 0, ctx->stream, un, \
                       (const T *)d_src, srcStride, (T *)d_rec, recStride, d_quant, d_nz, nunits, fs1, fs2, wrap, 7, 12 - inc)
    if (bytes_per_sample == 1) {
        if (size == 32) SVT_ENC_LAUNCH(32, uint8_t);
        else if (size == 16) SVT_ENC_LAUNCH(16, uint8_t);
        else if (size == 8) SVT_ENC_LAUNCH(8, uint8_t);
        else SVT_ENC_LAUNCH(4, uint8_t);
    } else {
        if (size == 32) SVT_ENC_LAUNCH(32, uint16_t);
        else if (size == 16) SVT_ENC_LAUNCH(16, uint16_t);
        else if (size == 8) SVT_ENC_LAUNCH(8, uint16_t);
        else SVT_ENC_LAUNCH(4, uint16_t);
    }
#undef SVT_ENC_LAUNCH
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
