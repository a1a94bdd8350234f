/*
 * Intra prediction kernels (8- and 16-bit): vertical / horizontal (luma with edge filter, chroma
 * without), DC (luma with edge filter), planar, angular 2 / 18 / 34 fast paths and the generic
 * vertical / horizontal angular kernels with intraPredAngle.
 * Reference: the 24 C kernels of C_DEFAULT/EbIntraPrediction_C.c:15-1051 behind the tables
 * IntraVerticalLuma_funcPtrArray ... IntraAngHorizontal_funcPtrArray (Codec/EbIntraPrediction.h:470-648).
 *
 * Batched form: block b reads ref + b*ref_pitch (samples) and writes pred + b*size*size;
 * one thread per predicted sample (HBM-bound: 1 write per sample, references hit in L1).
 * Reference-sample layout (size N): [0..2N) left + bottom-left, [2N] top-left, [2N+1..4N] top + top-right;
 * for the two generic angular kernels `main_offset` selects refSampMain inside the block's array.
 */
#include "leaf_util.h"

enum { IM_VERT_LUMA = 0, IM_VERT_CHROMA, IM_HOR_LUMA, IM_HOR_CHROMA, IM_DC_LUMA, IM_DC_CHROMA, IM_PLANAR,
       IM_ANG34, IM_ANG18, IM_ANG2, IM_ANG_VERT, IM_ANG_HOR, IM_COUNT };

template <typename T>
__global__ __launch_bounds__(256) void k_intra(int mode, int N, int skip, int angle, const T *__restrict__ refs,
                                               uint32_t ref_pitch, int main_offset, T *__restrict__ pred,
                                               uint32_t pred_stride, uint32_t pred_pitch, uint32_t nblocks)
{
    const int maxv = sizeof(T) == 1 ? 255 : 1023;
    const int lg = 31 - __clz(N);
    const uint32_t per = (uint32_t)N * N;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nblocks * per; i += gridDim.x * blockDim.x) {
        const uint32_t b = i / per, e = i - b * per;
        const int y = (int)(e >> lg), x = (int)(e & (N - 1));
        if (skip && (y & 1))
            continue;
        const T *r = refs + (size_t)b * ref_pitch;
        const int L = 0, TL = 2 * N, Tp = 2 * N + 1;
        int v = 0;
        switch (mode) {
        case IM_VERT_LUMA:
        case IM_VERT_CHROMA:
            v = r[Tp + x];
            if (mode == IM_VERT_LUMA && N < 32 && x == 0) {
                v = v + (((int)r[L + y] - (int)r[TL]) >> 1);
                v = v < 0 ? 0 : v > maxv ? maxv : v;
            }
            break;
        case IM_HOR_LUMA:
        case IM_HOR_CHROMA:
            v = r[L + y];
            if (mode == IM_HOR_LUMA && N < 32 && y == 0) {
                v = v + (((int)r[TL + x + 1] - (int)r[TL]) >> 1);
                v = v < 0 ? 0 : v > maxv ? maxv : v;
            }
            break;
        case IM_DC_LUMA:
        case IM_DC_CHROMA: {
            uint32_t sum = 0;
            for (int k = 0; k < N; k++)
                sum += (uint32_t)r[Tp + k] + (uint32_t)r[L + k];
            const int dc = (int)(T)((sum + (uint32_t)N) >> (lg + 1));
            v = dc;
            if (mode == IM_DC_LUMA && N < 32) {
                if (x == 0 && y == 0)
                    v = (int)(T)(((int)r[L] + (int)r[Tp] + (dc << 1) + 2) >> 2);
                else if (y == 0)
                    v = (int)(T)(((int)r[Tp + x] + 3 * dc + 2) >> 2);
                else if (x == 0)
                    v = (int)(T)(((int)r[L + y] + 3 * dc + 2) >> 2);
            }
            break;
        }
        case IM_PLANAR:
            v = (int)(T)(((uint32_t)(N - 1 - x) * r[L + y] + (uint32_t)(x + 1) * r[Tp + N] +
                          (uint32_t)(N - 1 - y) * r[Tp + x] + (uint32_t)(y + 1) * r[L + N] + (uint32_t)N) >> (lg + 1));
            break;
        case IM_ANG34: v = r[Tp + y + x + 1]; break;
        case IM_ANG18: v = r[TL - y + x]; break;
        case IM_ANG2: v = r[L + y + x + 1]; break;
        case IM_ANG_VERT: {
            const T *m = r + main_offset + 1;
            const int ds = (y + 1) * angle, di = ds >> 5, df = ds & 31;
            v = (int)(T)(((32 - df) * (int)m[di + x] + df * (int)m[di + x + 1] + 16) >> 5);
            break;
        }
        default: { /* IM_ANG_HOR */
            const T *m = r + main_offset + 1;
            const int ds = (x + 1) * angle, di = ds >> 5, df = ds & 31;
            v = (int)(T)(((32 - df) * (int)m[di + y] + df * (int)m[di + y + 1] + 16) >> 5);
            break;
        }
        }
        pred[(size_t)b * pred_pitch + (size_t)y * pred_stride + x] = (T)v;
    }
}

int svt_amd_launch_intra(hipStream_t st, int mode, int bps, int size, int skip, int angle, const void *d_refs,
                         uint32_t ref_pitch, int main_offset, void *d_pred, uint32_t pred_stride, uint32_t pred_pitch,
                         uint32_t n)
{
    if (mode < 0 || mode >= IM_COUNT || (bps != 1 && bps != 2) || !n ||
        !(size == 4 || size == 8 || size == 16 || size == 32 || size == 64))
        return SVT_AMD_ERR_BAD_PARAM;
    const uint32_t total = n * (uint32_t)size * size;
    const dim3 grid((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (bps == 1)
        hipLaunchKernelGGL(k_intra<uint8_t>, grid, dim3(256), 0, st, mode, size, skip, angle, (const uint8_t *)d_refs,
                           ref_pitch, main_offset, (uint8_t *)d_pred, pred_stride, pred_pitch, n);
    else
        hipLaunchKernelGGL(k_intra<uint16_t>, grid, dim3(256), 0, st, mode, size, skip, angle, (const uint16_t *)d_refs,
                           ref_pitch, main_offset, (uint16_t *)d_pred, pred_stride, pred_pitch, n);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

extern "C" int svt_amd_intra_pred_batch(SvtAmdContext *ctx, int mode, int bytes_per_sample, int size, int skip,
                                        int32_t intraPredAngle, const void *d_refs, uint32_t ref_pitch,
                                        int32_t main_offset, void *d_pred, uint32_t nblocks)
{
    if (!ctx || !d_refs || !d_pred)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    return svt_amd_launch_intra(ctx->stream, mode, bytes_per_sample, size, skip, intraPredAngle, d_refs, ref_pitch,
                                main_offset, d_pred, (uint32_t)size, (uint32_t)size * size, nblocks);
}

/* ---- LEAF wrappers (host pointers, reference signatures) ---- */
static void intra_leaf(int mode, int bps, uint32_t size, const void *ref, void *pred, uint32_t stride, int skip, int angle)
{
    const bool ang = mode == IM_ANG_VERT || mode == IM_ANG_HOR;
    /* generic angular kernels index refSampMain[1 - size .. 2*size + 2] */
    const size_t lead = ang ? size : 0, count = ang ? 3 * (size_t)size + 4 : 4 * (size_t)size + 2;
    DBuf r((const uint8_t *)ref - lead * bps, count * bps), p(pred, span(stride, size, size) * bps);
    if (!(r.ok && p.ok))
        return;
    if (svt_amd_launch_intra(0, mode, bps, (int)size, skip, angle, r.d, 0, (int)lead, p.d, stride, 0, 1) || !finish("intra"))
        return;
    p.download(pred, span(stride, size, size) * bps);
}
#define INTRA_LEAF(name, T, mode)                                                                                  \
    extern "C" void svt_amd_##name(const uint32_t size, T *refSamples, T *predictionPtr,                           \
                                   const uint32_t predictionBufferStride, const uint8_t skip)                      \
    {                                                                                                              \
        intra_leaf(mode, (int)sizeof(T), size, refSamples, predictionPtr, predictionBufferStride, skip, 0);        \
    }
#define INTRA_ANG_LEAF(name, T, mode)                                                                              \
    extern "C" void svt_amd_##name(uint32_t size, T *refSampMain, T *predictionPtr, uint32_t predictionBufferStride, \
                                   const uint8_t skip, int32_t intraPredAngle)                                     \
    {                                                                                                              \
        intra_leaf(mode, (int)sizeof(T), size, refSampMain, predictionPtr, predictionBufferStride, skip, intraPredAngle); \
    }
INTRA_LEAF(IntraModeVerticalLuma, uint8_t, IM_VERT_LUMA)
INTRA_LEAF(IntraModeVerticalLuma16bit, uint16_t, IM_VERT_LUMA)
INTRA_LEAF(IntraModeVerticalChroma, uint8_t, IM_VERT_CHROMA)
INTRA_LEAF(IntraModeVerticalChroma16bit, uint16_t, IM_VERT_CHROMA)
INTRA_LEAF(IntraModeHorizontalLuma, uint8_t, IM_HOR_LUMA)
INTRA_LEAF(IntraModeHorizontalLuma16bit, uint16_t, IM_HOR_LUMA)
INTRA_LEAF(IntraModeHorizontalChroma, uint8_t, IM_HOR_CHROMA)
INTRA_LEAF(IntraModeHorizontalChroma16bit, uint16_t, IM_HOR_CHROMA)
INTRA_LEAF(IntraModeDCLuma, uint8_t, IM_DC_LUMA)
INTRA_LEAF(IntraModeDCLuma16bit, uint16_t, IM_DC_LUMA)
INTRA_LEAF(IntraModeDCChroma, uint8_t, IM_DC_CHROMA)
INTRA_LEAF(IntraModeDCChroma16bit, uint16_t, IM_DC_CHROMA)
INTRA_LEAF(IntraModePlanar, uint8_t, IM_PLANAR)
INTRA_LEAF(IntraModePlanar16bit, uint16_t, IM_PLANAR)
INTRA_LEAF(IntraModeAngular_34, uint8_t, IM_ANG34)
INTRA_LEAF(IntraModeAngular16bit_34, uint16_t, IM_ANG34)
INTRA_LEAF(IntraModeAngular_18, uint8_t, IM_ANG18)
INTRA_LEAF(IntraModeAngular16bit_18, uint16_t, IM_ANG18)
INTRA_LEAF(IntraModeAngular_2, uint8_t, IM_ANG2)
INTRA_LEAF(IntraModeAngular16bit_2, uint16_t, IM_ANG2)
INTRA_ANG_LEAF(IntraModeAngular_Vertical_Kernel, uint8_t, IM_ANG_VERT)
INTRA_ANG_LEAF(IntraModeAngular16bit_Vertical_Kernel, uint16_t, IM_ANG_VERT)
INTRA_ANG_LEAF(IntraModeAngular_Horizontal_Kernel, uint8_t, IM_ANG_HOR)
INTRA_ANG_LEAF(IntraModeAngular16bit_Horizontal_Kernel, uint16_t, IM_ANG_HOR)
