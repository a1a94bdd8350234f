/*
 * LEAF layer of the C-ABI: table-slot replacements on HOST pointers.
 * Each call stages its operands in device memory, runs one small HIP kernel and
 * copies the result back - per-call differential parity against the C_DEFAULT
 * symbols, never the fast path (that is the batched layer, me_kernels.hip).
 * The reference symbol each entry point replaces is cited in include/svt_hevc_amd.h.
 * Failure policy: a HIP error aborts via svt_amd_last_error() + SIGABRT-free
 * poison results (all-ones) - there is no CPU fallback.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "svt_amd_internal.h"

typedef uint32_t __attribute__((aligned(1))) u32u;
__device__ __forceinline__ uint32_t l_absd(uint32_t a, uint32_t b) { return a > b ? a - b : b - a; }

#include "leaf_util.h"

/* -------- kernels (one workgroup of 256 threads) -------- */

__global__ void k_sad_nxm(const uint8_t *src, uint32_t ss, const uint8_t *ref, uint32_t rs, uint32_t h, uint32_t w,
                          uint32_t *out)
{
    __shared__ uint32_t acc;
    if (threadIdx.x == 0)
        acc = 0;
    __syncthreads();
    uint32_t s = 0;
    for (uint32_t i = threadIdx.x; i < h * w; i += blockDim.x) {
        const uint32_t y = i / w, x = i - y * w;
        s += l_absd(src[y * ss + x], ref[y * rs + x]);
    }
    atomicAdd(&acc, s);
    __syncthreads();
    if (threadIdx.x == 0)
        *out = acc;
}

__global__ void k_sad_avg(const uint8_t *src, uint32_t ss, const uint8_t *r1, uint32_t s1, const uint8_t *r2,
                          uint32_t s2, uint32_t h, uint32_t w, uint32_t *out)
{
    __shared__ uint32_t acc;
    if (threadIdx.x == 0)
        acc = 0;
    __syncthreads();
    uint32_t s = 0;
    for (uint32_t i = threadIdx.x; i < h * w; i += blockDim.x) {
        const uint32_t y = i / w, x = i - y * w;
        const uint32_t a = ((uint32_t)r1[y * s1 + x] + r2[y * s2 + x] + 1) >> 1;
        s += l_absd(src[y * ss + x], a);
    }
    atomicAdd(&acc, s);
    __syncthreads();
    if (threadIdx.x == 0)
        *out = acc;
}

/* raster search, first minimum wins: key = sad << 32 | index */
__global__ void k_sad_loop(const uint8_t *src, uint32_t ss, const uint8_t *ref, uint32_t rs, uint32_t h, uint32_t w,
                           uint32_t raw, int saw, int sah, unsigned long long *out)
{
    __shared__ unsigned long long best;
    if (threadIdx.x == 0)
        best = ~0ull;
    __syncthreads();
    unsigned long long mine = ~0ull;
    for (int p = threadIdx.x; p < saw * sah; p += blockDim.x) {
        const int sy = p / saw, sx = p - sy * saw;
        const uint8_t *r = ref + (size_t)sy * raw + sx;
        uint32_t s = 0;
        for (uint32_t y = 0; y < h; y++)
            for (uint32_t x = 0; x < w; x++)
                s += l_absd(src[y * ss + x], r[y * rs + x]);
        const unsigned long long k = ((unsigned long long)s << 32) | (uint32_t)p;
        mine = k < mine ? k : mine;
    }
    atomicMin(&best, mine);
    __syncthreads();
    if (threadIdx.x == 0)
        *out = best;
}

/* 8 positions x four 8x8 blocks, even rows; serial update by thread 0 as in the C code */
__global__ void k_eight_8x8_16x16(const uint8_t *src, uint32_t ss, const uint8_t *ref, uint32_t rs, uint32_t *bs8,
                                  uint32_t *bm8, uint32_t *bs16, uint32_t *bm16, uint32_t mv, uint16_t *s16out,
                                  int npos)
{
    __shared__ uint32_t s[8][4];
    const int t = threadIdx.x;
    if (t < npos * 4) {
        const int i = t >> 2, k = t & 3;
        const uint8_t *a = src + (k >> 1) * 8 * ss + (k & 1) * 8, *b = ref + i + (k >> 1) * 8 * rs + (k & 1) * 8;
        uint32_t v = 0;
        for (int y = 0; y < 8; y += 2)
            for (int x = 0; x < 8; x += 4)
                v = __builtin_amdgcn_sad_u8(*(const u32u *)(a + y * ss + x), *(const u32u *)(b + y * rs + x), v);
        s[i][k] = v;
    }
    __syncthreads();
    if (t == 0) {
        const int16_t mx = (int16_t)(mv & 0xffff), my = (int16_t)(mv >> 16);
        for (int i = 0; i < npos; i++) {
            const uint32_t here = ((uint32_t)(uint16_t)my << 16) | (uint16_t)(int16_t)(mx + (int16_t)i * 4);
            if (npos == 8) { /* GetEightHorizontalSearchPointResults_8x8_16x16_PU */
                for (int k = 0; k < 4; k++)
                    if (2 * s[i][k] < bs8[k])
                        bs8[k] = 2 * s[i][k], bm8[k] = here;
                const uint16_t v = (uint16_t)(s[i][0] + s[i][1] + s[i][2] + s[i][3]);
                s16out[i] = v;
                if ((uint32_t)(2 * v) < bs16[0])
                    bs16[0] = 2 * v, bm16[0] = here;
            } else { /* SadCalculation_8x8_16x16: one position, u32 output */
                unsigned long long tot = 0;
                for (int k = 0; k < 4; k++) {
                    const unsigned long long v = (unsigned long long)s[0][k] << 1;
                    if (v < bs8[k])
                        bs8[k] = (uint32_t)v, bm8[k] = mv;
                    tot += v;
                }
                if (tot < bs16[0])
                    bs16[0] = (uint32_t)tot, bm16[0] = mv;
                *(uint32_t *)s16out = (uint32_t)tot;
            }
        }
    }
}

__global__ void k_tree_32_64(const uint16_t *s16_u16, const uint32_t *s16_u32, uint32_t *bs32, uint32_t *bs64,
                             uint32_t *bm32, uint32_t *bm64, uint32_t mv, int npos)
{
    if (threadIdx.x != 0)
        return;
    const int16_t mx = (int16_t)(mv & 0xffff), my = (int16_t)(mv >> 16);
    for (int i = 0; i < npos; i++) {
        uint32_t s64 = 0;
        const uint32_t here = npos == 8 ? (((uint32_t)(uint16_t)my << 16) | (uint16_t)(int16_t)(mx + (int16_t)i * 4)) : mv;
        for (int q = 0; q < 4; q++) {
            uint32_t s32 = 0;
            for (int k = 0; k < 4; k++)
                s32 += npos == 8 ? s16_u16[(4 * q + k) * 8 + i] : s16_u32[4 * q + k];
            const uint32_t c = npos == 8 ? 2 * s32 : s32;
            if (c < bs32[q])
                bs32[q] = c, bm32[q] = here;
            s64 += s32;
        }
        const uint32_t c64 = npos == 8 ? 2 * s64 : s64;
        if (npos == 8 ? (c64 <= bs64[0]) : (c64 < bs64[0]))
            bs64[0] = c64, bm64[0] = here;
    }
}

__constant__ int8_t c_avc[4][4] = {{0, 0, 0, 0}, {-1, 25, 9, -1}, {-2, 18, 18, -2}, {-1, 9, 25, -1}};
__global__ void k_avc_filter(const uint8_t *ref, uint32_t rs, uint8_t *dst, uint32_t ds, uint32_t w, uint32_t h,
                             uint32_t frac, int vertical)
{
    const int step = vertical ? (int)rs : 1;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / w, x = i - y * w;
        const uint8_t *p = ref + (size_t)y * rs + x;
        int v = p[-step] * c_avc[frac][0] + p[0] * c_avc[frac][1] + p[step] * c_avc[frac][2] +
                p[2 * step] * c_avc[frac][3] + 16;
        v >>= 5;
        dst[(size_t)y * ds + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

__global__ void k_average(const uint8_t *a, uint32_t as, const uint8_t *b, uint32_t bs, uint8_t *d, uint32_t ds,
                          uint32_t w, uint32_t h)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / w, x = i - y * w;
        d[(size_t)y * ds + x] = (uint8_t)(((uint32_t)a[(size_t)y * as + x] + b[(size_t)y * bs + x] + 1) >> 1);
    }
}

__global__ void k_sse(const uint8_t *a, uint32_t as, const uint8_t *b, uint32_t bs, uint32_t w, uint32_t h,
                      unsigned long long *out)
{
    __shared__ unsigned long long acc;
    if (threadIdx.x == 0)
        acc = 0;
    __syncthreads();
    unsigned long long s = 0;
    for (uint32_t i = threadIdx.x; i < w * h; i += blockDim.x) {
        const uint32_t y = i / w, x = i - y * w;
        const long long e = (long long)a[(size_t)y * as + x] - b[(size_t)y * bs + x];
        s += (unsigned long long)(e * e);
    }
    atomicAdd(&acc, s);
    __syncthreads();
    if (threadIdx.x == 0)
        *out = acc;
}

__global__ void k_decimate(const uint8_t *in, uint32_t is, uint32_t w, uint32_t h, uint8_t *out, uint32_t os,
                           uint32_t step)
{
    const uint32_t ow = (w + step - 1) / step, oh = (h + step - 1) / step;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ow * oh; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / ow, x = i - y * ow;
        out[(size_t)y * os + x] = in[(size_t)y * step * is + x * step];
    }
}

/* -------- entry points -------- */

extern "C" uint32_t svt_amd_NxMSadKernel(const uint8_t *src, uint32_t srcStride, const uint8_t *ref,
                                         uint32_t refStride, uint32_t height, uint32_t width)
{
    DBuf a(src, span(srcStride, width, height)), b(ref, span(refStride, width, height)), o(nullptr, 8, false);
    uint32_t r = 0xffffffffu;
    if (!(a.ok && b.ok && o.ok))
        return r;
    hipLaunchKernelGGL(k_sad_nxm, dim3(1), dim3(256), 0, 0, a.d, srcStride, b.d, refStride, height, width, (uint32_t *)o.d);
    if (finish("NxMSadKernel"))
        o.download(&r, 4);
    return r;
}

extern "C" uint32_t svt_amd_NxMSadAveragingKernel(const uint8_t *src, uint32_t srcStride, const uint8_t *ref1,
                                                  uint32_t ref1Stride, const uint8_t *ref2, uint32_t ref2Stride,
                                                  uint32_t height, uint32_t width)
{
    DBuf a(src, span(srcStride, width, height)), b(ref1, span(ref1Stride, width, height)),
        c(ref2, span(ref2Stride, width, height)), o(nullptr, 8, false);
    uint32_t r = 0xffffffffu;
    if (!(a.ok && b.ok && c.ok && o.ok))
        return r;
    hipLaunchKernelGGL(k_sad_avg, dim3(1), dim3(256), 0, 0, a.d, srcStride, b.d, ref1Stride, c.d, ref2Stride, height,
                       width, (uint32_t *)o.d);
    if (finish("NxMSadAveragingKernel"))
        o.download(&r, 4);
    return r;
}

extern "C" void svt_amd_SadLoopKernel(const uint8_t *src, uint32_t srcStride, const uint8_t *ref, uint32_t refStride,
                                      uint32_t height, uint32_t width, uint64_t *bestSad, int16_t *xSearchCenter,
                                      int16_t *ySearchCenter, uint32_t srcStrideRaw, int16_t searchAreaWidth,
                                      int16_t searchAreaHeight)
{
    *bestSad = 0xffffff;
    if (searchAreaWidth <= 0 || searchAreaHeight <= 0)
        return;
    const size_t refbytes = (size_t)(searchAreaHeight - 1) * srcStrideRaw + span(refStride, width, height) + searchAreaWidth;
    DBuf a(src, span(srcStride, width, height)), b(ref, refbytes), o(nullptr, 8, false);
    if (!(a.ok && b.ok && o.ok))
        return;
    hipLaunchKernelGGL(k_sad_loop, dim3(1), dim3(256), 0, 0, a.d, srcStride, b.d, refStride, height, width,
                       srcStrideRaw, (int)searchAreaWidth, (int)searchAreaHeight, (unsigned long long *)o.d);
    unsigned long long k = ~0ull;
    if (!finish("SadLoopKernel") || !o.download(&k, 8))
        return;
    if ((k >> 32) < 0xffffff) {
        const int p = (int)(uint32_t)k;
        *bestSad = k >> 32;
        *xSearchCenter = (int16_t)(p % searchAreaWidth);
        *ySearchCenter = (int16_t)(p / searchAreaWidth);
    }
}

static void eight_or_one(const uint8_t *src, uint32_t ss, const uint8_t *ref, uint32_t rs, uint32_t *bs8,
                         uint32_t *bm8, uint32_t *bs16, uint32_t *bm16, uint32_t mv, void *s16, int npos)
{
    DBuf a(src, span(ss, 16, 16)), b(ref, span(rs, 16, 16) + (size_t)npos), d8(bs8, 16), m8(bm8, 16), d16(bs16, 4),
        m16(bm16, 4), so(nullptr, 16, false);
    if (!(a.ok && b.ok && d8.ok && m8.ok && d16.ok && m16.ok && so.ok))
        return;
    hipLaunchKernelGGL(k_eight_8x8_16x16, dim3(1), dim3(64), 0, 0, a.d, ss, b.d, rs, (uint32_t *)d8.d, (uint32_t *)m8.d,
                       (uint32_t *)d16.d, (uint32_t *)m16.d, mv, (uint16_t *)so.d, npos);
    if (!finish("SAD 8x8/16x16"))
        return;
    d8.download(bs8, 16);
    m8.download(bm8, 16);
    d16.download(bs16, 4);
    m16.download(bm16, 4);
    so.download(s16, npos == 8 ? 16 : 4);
}

extern "C" void svt_amd_GetEightHorizontalSearchPointResults_8x8_16x16_PU(
    const uint8_t *src, uint32_t srcStride, const uint8_t *ref, uint32_t refStride, uint32_t *pBestSad8x8,
    uint32_t *pBestMV8x8, uint32_t *pBestSad16x16, uint32_t *pBestMV16x16, uint32_t mv, uint16_t *pSad16x16)
{
    eight_or_one(src, srcStride, ref, refStride, pBestSad8x8, pBestMV8x8, pBestSad16x16, pBestMV16x16, mv, pSad16x16, 8);
}

extern "C" void svt_amd_SadCalculation_8x8_16x16(const uint8_t *src, uint32_t srcStride, const uint8_t *ref,
                                                 uint32_t refStride, uint32_t *pBestSad8x8, uint32_t *pBestSad16x16,
                                                 uint32_t *pBestMV8x8, uint32_t *pBestMV16x16, uint32_t mv,
                                                 uint32_t *pSad16x16)
{
    eight_or_one(src, srcStride, ref, refStride, pBestSad8x8, pBestMV8x8, pBestSad16x16, pBestMV16x16, mv, pSad16x16, 1);
}

static void tree(const void *s16, size_t s16bytes, uint32_t *bs32, uint32_t *bs64, uint32_t *bm32, uint32_t *bm64,
                 uint32_t mv, int npos)
{
    DBuf s(s16, s16bytes), a(bs32, 16), b(bs64, 4), c(bm32, 16), d(bm64, 4);
    if (!(s.ok && a.ok && b.ok && c.ok && d.ok))
        return;
    hipLaunchKernelGGL(k_tree_32_64, dim3(1), dim3(64), 0, 0, (const uint16_t *)s.d, (const uint32_t *)s.d,
                       (uint32_t *)a.d, (uint32_t *)b.d, (uint32_t *)c.d, (uint32_t *)d.d, mv, npos);
    if (!finish("SAD 32x32/64x64"))
        return;
    a.download(bs32, 16);
    b.download(bs64, 4);
    c.download(bm32, 16);
    d.download(bm64, 4);
}

extern "C" void svt_amd_GetEightHorizontalSearchPointResults_32x32_64x64(const uint16_t *pSad16x16,
                                                                         uint32_t *pBestSad32x32,
                                                                         uint32_t *pBestSad64x64,
                                                                         uint32_t *pBestMV32x32,
                                                                         uint32_t *pBestMV64x64, uint32_t mv)
{
    tree(pSad16x16, 16 * 8 * 2, pBestSad32x32, pBestSad64x64, pBestMV32x32, pBestMV64x64, mv, 8);
}

extern "C" void svt_amd_SadCalculation_32x32_64x64(const uint32_t *pSad16x16, uint32_t *pBestSad32x32,
                                                   uint32_t *pBestSad64x64, uint32_t *pBestMV32x32,
                                                   uint32_t *pBestMV64x64, uint32_t mv)
{
    tree(pSad16x16, 16 * 4, pBestSad32x32, pBestSad64x64, pBestMV32x32, pBestMV64x64, mv, 1);
}

static void avc(const uint8_t *refPic, uint32_t srcStride, uint8_t *dst, uint32_t dstStride, uint32_t w, uint32_t h,
                uint32_t frac, int vertical)
{
    /* taps reach 1 sample before and 2 after along the filtered axis */
    const uint8_t *base = vertical ? refPic - srcStride : refPic - 1;
    const size_t bytes = vertical ? span(srcStride, w, h + 3) : span(srcStride, w + 3, h);
    DBuf a(base, bytes), d(dst, span(dstStride, w, h));
    if (!(a.ok && d.ok))
        return;
    hipLaunchKernelGGL(k_avc_filter, dim3(64), dim3(256), 0, 0, a.d + (vertical ? srcStride : 1), srcStride, d.d,
                       dstStride, w, h, frac, vertical);
    if (finish("AvcStyleLumaInterpolationFilter"))
        d.download(dst, span(dstStride, w, h));
}

extern "C" void svt_amd_AvcStyleLumaInterpolationFilterHorizontal(const uint8_t *refPic, uint32_t srcStride,
                                                                  uint8_t *dst, uint32_t dstStride, uint32_t puWidth,
                                                                  uint32_t puHeight, uint8_t *tempBuf, uint32_t fracPos)
{
    (void)tempBuf;
    avc(refPic, srcStride, dst, dstStride, puWidth, puHeight, fracPos, 0);
}

extern "C" void svt_amd_AvcStyleLumaInterpolationFilterVertical(const uint8_t *refPic, uint32_t srcStride,
                                                                uint8_t *dst, uint32_t dstStride, uint32_t puWidth,
                                                                uint32_t puHeight, uint8_t *tempBuf, uint32_t fracPos)
{
    (void)tempBuf;
    avc(refPic, srcStride, dst, dstStride, puWidth, puHeight, fracPos, 1);
}

extern "C" void svt_amd_PictureAverageKernel(const uint8_t *src0, uint32_t src0Stride, const uint8_t *src1,
                                             uint32_t src1Stride, uint8_t *dst, uint32_t dstStride,
                                             uint32_t areaWidth, uint32_t areaHeight)
{
    DBuf a(src0, span(src0Stride, areaWidth, areaHeight)), b(src1, span(src1Stride, areaWidth, areaHeight)),
        d(dst, span(dstStride, areaWidth, areaHeight));
    if (!(a.ok && b.ok && d.ok))
        return;
    hipLaunchKernelGGL(k_average, dim3(64), dim3(256), 0, 0, a.d, src0Stride, b.d, src1Stride, d.d, dstStride,
                       areaWidth, areaHeight);
    if (finish("PictureAverageKernel"))
        d.download(dst, span(dstStride, areaWidth, areaHeight));
}

extern "C" uint64_t svt_amd_SpatialFullDistortionKernel(const uint8_t *input, uint32_t inputStride,
                                                        const uint8_t *recon, uint32_t reconStride,
                                                        uint32_t areaWidth, uint32_t areaHeight)
{
    DBuf a(input, span(inputStride, areaWidth, areaHeight)), b(recon, span(reconStride, areaWidth, areaHeight)),
        o(nullptr, 8, false);
    uint64_t r = ~0ull;
    if (!(a.ok && b.ok && o.ok))
        return r;
    hipLaunchKernelGGL(k_sse, dim3(1), dim3(256), 0, 0, a.d, inputStride, b.d, reconStride, areaWidth, areaHeight,
                       (unsigned long long *)o.d);
    if (finish("SpatialFullDistortionKernel"))
        o.download(&r, 8);
    return r;
}

extern "C" void svt_amd_Decimation2D(const uint8_t *inputSamples, uint32_t inputStride, uint32_t inputAreaWidth,
                                     uint32_t inputAreaHeight, uint8_t *decimSamples, uint32_t decimStride,
                                     uint32_t decimStep)
{
    const uint32_t ow = (inputAreaWidth + decimStep - 1) / decimStep, oh = (inputAreaHeight + decimStep - 1) / decimStep;
    DBuf a(inputSamples, span(inputStride, inputAreaWidth, inputAreaHeight)), d(decimSamples, span(decimStride, ow, oh));
    if (!(a.ok && d.ok))
        return;
    hipLaunchKernelGGL(k_decimate, dim3(64), dim3(256), 0, 0, a.d, inputStride, inputAreaWidth, inputAreaHeight, d.d,
                       decimStride, decimStep);
    if (finish("Decimation2D"))
        d.download(decimSamples, span(decimStride, ow, oh));
}
