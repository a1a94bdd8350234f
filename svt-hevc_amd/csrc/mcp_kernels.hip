/*
 * HEVC motion-compensation interpolation (EncDec row "Inter2Nx2NPuPredictionHevc ... MCP leaf set").
 *
 * One kernel covers the reference's ~150 leaf functions (C_DEFAULT/EbMcp_C.c; tables Codec/EbMcpTables.c:14-745):
 * luma 8-tap at the 16 quarter-pel positions, chroma 4-tap at the 64 eighth-pel positions, clipped uni-prediction
 * output or biased int16 raw output for bi-prediction, 8- and 10-bit samples - they are all the H.265 8.5.3.3.3
 * separable filter with the fixed-point conventions listed in oracle/svt_oracle_mcp.c.
 * One workgroup per prediction block: the reference window goes to LDS once, 2-D positions run the horizontal pass into
 * a second LDS array (int16, h + taps - 1 rows), barrier, vertical pass; four outputs per thread and pass.
 * Plus BiPredClipping(16bit): average of two raw blocks.
 * Bound: HBM for the copy / 1-D positions (1-2 bytes in, 1-2 out per sample), ALU-light otherwise.
 */
#include "leaf_util.h"

struct McpBlock { int32_t ref_off, dst_off; uint16_t w, h; uint8_t fx, fy, pad[2]; }; /* = SvtAmdMcpBlock */
struct BiBlock { int32_t l0_off, l1_off, dst_off; uint16_t w, h; };                     /* = SvtAmdBiPredBlock */

__device__ __constant__ int8_t c_luma_taps[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0},
                                                    {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
__device__ __constant__ int8_t c_chroma_taps[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                                      {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

/* One workgroup per block.  The reference window is staged in LDS once (every sample of it is read from memory exactly
 * once); each pass then produces four neighbouring outputs per thread from one run of taps+3 LDS samples (2.75 LDS
 * reads per output instead of 8 memory reads).  Block widths and heights are multiples of 4. */
#define MCP_NT 64 /* one wave per block: the blocks are small (16x16 is typical) and independent, barriers stay inside a wave */
/* MAXD: largest block width / height of the launch - it sizes the two LDS arrays and with them the number of blocks a
 * CU keeps in flight (16: 1.5 KB per block, 64: 15 KB) */
/* TS: sample type of the reference plane.  TS = uint16_t with T = uint8_t is the mode decision of a 10-bit encode: it predicts from
 * the 8 most significant bits of the 16-bit reference pictures (UnPackReferenceBlock -> Extract8BitdataSafeSub = sample >> 2,
 * Codec/EbInterPrediction.c:414-457, C_DEFAULT/EbPackUnPack_C.c:203-225), so the window is narrowed while it is staged. */
template <typename T, int MAXD, typename TS = T>
__global__ __launch_bounds__(MCP_NT) void k_mcp(const TS *__restrict__ ref, int rstride, void *__restrict__ dst, int dstride,
                                             const McpBlock *__restrict__ blocks, int chroma, int out_raw)
{
    constexpr int MSB_SHIFT = sizeof(TS) > sizeof(T) ? 2 : 0;
    constexpr int WP = MAXD + 8;                    /* window pitch */
    __shared__ T win[(MAXD + 7 + 4) * WP];          /* + 4 rows: the four-output groups of the last rows read past the window */
    __shared__ int16_t tmp[(MAXD + 7 + 4) * MAXD];
    const McpBlock b = blocks[blockIdx.x];
    const int w = b.w, h = b.h, fx = b.fx, fy = b.fy, t = threadIdx.x;
    const int ntaps = chroma ? 4 : 8, first = chroma ? -1 : -3;
    const int8_t *tx = chroma ? c_chroma_taps[fx & 7] : c_luma_taps[fx & 3];
    const int8_t *ty = chroma ? c_chroma_taps[fy & 7] : c_luma_taps[fy & 3];
    constexpr int s1 = sizeof(T) == 1 ? 0 : 2, maxv = sizeof(T) == 1 ? 255 : 1023;
    const int B = (sizeof(T) == 2 || !chroma) ? 8192 : 0;
    const TS *r0 = ref + b.ref_off;
    const int ds = out_raw ? w : dstride;
    /* only the taps the reference's own functions read (the 7-tap quarter positions never touch the 8th sample) */
    const int kx0 = fx ? ((!chroma && fx == 3) ? 1 : 0) : 0, kx1 = fx ? ((!chroma && fx == 1) ? 7 : ntaps) : 1;
    const int ky0 = fy ? ((!chroma && fy == 3) ? 1 : 0) : 0, ky1 = fy ? ((!chroma && fy == 1) ? 7 : ntaps) : 1;
    /* window: columns [ox, ox + cols), rows [oy, oy + rows) relative to the block */
    const int ox = fx ? first + kx0 : 0, cols = w + (kx1 - kx0 - 1), oy = fy ? first + ky0 : 0, rows = h + (ky1 - ky0 - 1);
    {
        const uint32_t rc = (1u << 20) / (uint32_t)cols + 1u; /* i / cols for i < 2^13 by reciprocal multiplication */
        for (int i = t; i < rows * cols; i += MCP_NT) {
            const int j = (int)(((uint32_t)i * rc) >> 20), x = i - j * cols;
            win[j * WP + x] = (T)(r0[(ptrdiff_t)(j + oy) * rstride + x + ox] >> MSB_SHIFT);
        }
    }
    __syncthreads();
    auto put = [&](int x, int y, int v) {
        if (out_raw)
            ((int16_t *)dst)[b.dst_off + y * ds + x] = (int16_t)v;
        else
            ((T *)dst)[b.dst_off + (ptrdiff_t)y * ds + x] = (T)v;
    };
    const int wq = (w + 3) >> 2, hq = (h + 3) >> 2; /* groups of four (the last one may be partial) */
    if (!fx && !fy) {
        for (int i = t; i < w * h; i += MCP_NT) {
            const int y = i / w, x = i - y * w, p = win[y * WP + x];
            put(x, y, out_raw ? (int16_t)((p << (6 - s1)) - B) : p);
        }
        return;
    }
    if (fx) { /* horizontal pass over every window row: four outputs per thread from one run of samples */
        const int nt = kx1 - kx0;
        const int lgq = 31 - __clz(wq), gpr = 1 << lgq; /* wq is 1, 2, 4, 8 or 16 for the power-of-two block widths ... */
        const bool pow2 = gpr == wq;
        for (int i = t; i < rows * wq; i += MCP_NT) {
            const int j = pow2 ? i >> lgq : i / wq, x0 = (i - j * wq) * 4; /* ... other widths take the division */
            int smp[11];
#pragma unroll
            for (int k = 0; k < 11; k++)
                smp[k] = k < nt + 3 ? (int)win[j * WP + x0 + k] : 0;
            int acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < nt) {
                    const int c = tx[kx0 + k];
                    acc[0] += c * smp[k], acc[1] += c * smp[k + 1], acc[2] += c * smp[k + 2], acc[3] += c * smp[k + 3];
                }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (x0 + q >= w)
                    break;
                if (fy)
                    tmp[j * w + x0 + q] = (int16_t)((acc[q] - (B << s1)) >> s1);
                else
                    put(x0 + q, j, out_raw ? ((acc[q] - (B << s1)) >> s1) : min(maxv, max(0, (acc[q] + 32) >> 6)));
            }
        }
        if (!fy)
            return;
        __syncthreads();
    }
    /* vertical pass: four outputs of one column per thread */
    const int nt = ky1 - ky0;
    const int lgw = 31 - __clz(w);
    const bool wpow2 = (1 << lgw) == w;
    for (int i = t; i < hq * w; i += MCP_NT) {
        const int yq = wpow2 ? i >> lgw : i / w, x = i - yq * w, y0 = yq * 4;
        int smp[11];
#pragma unroll
        for (int k = 0; k < 11; k++)
            smp[k] = k < nt + 3 ? (fx ? (int)tmp[(y0 + k) * w + x] : (int)win[(y0 + k) * WP + x]) : 0;
        int acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < nt) {
                const int c = ty[ky0 + k];
                acc[0] += c * smp[k], acc[1] += c * smp[k + 1], acc[2] += c * smp[k + 2], acc[3] += c * smp[k + 3];
            }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (y0 + q >= h)
                break;
            int v;
            if (fx)
                v = out_raw ? (acc[q] >> 6) : min(maxv, max(0, (acc[q] + (B << 6) + (1 << (11 - s1))) >> (12 - s1)));
            else
                v = out_raw ? ((acc[q] - (B << s1)) >> s1) : min(maxv, max(0, (acc[q] + 32) >> 6));
            put(x, y0 + q, v);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_bipred_clip(const int16_t *__restrict__ l0, const int16_t *__restrict__ l1,
                                                     T *__restrict__ dst, int dstride, const BiBlock *__restrict__ blocks,
                                                     int offset)
{
    const BiBlock b = blocks[blockIdx.x];
    const int w = b.w, h = b.h;
    for (int i = threadIdx.x; i < w * h; i += 256) {
        const int y = i / w, x = i - y * w;
        const int s = (int)l0[b.l0_off + i] + (int)l1[b.l1_off + i];
        dst[b.dst_off + (ptrdiff_t)y * dstride + x] =
            sizeof(T) == 1 ? (T)min(255, max(0, (s + offset) >> 7)) : (T)min(1023, max(0, (s + 16400) >> 5));
    }
}

template <typename T, typename TS = T>
static void launch_mcp(hipStream_t st, uint32_t nblocks, uint32_t max_dim, const TS *ref, int rstride, void *dst, int dstride,
                       const McpBlock *blocks, int chroma, int out_raw)
{
    if (max_dim && max_dim <= 16)
        hipLaunchKernelGGL((k_mcp<T, 16, TS>), dim3(nblocks), dim3(MCP_NT), 0, st, ref, rstride, dst, dstride, blocks, chroma, out_raw);
    else if (max_dim && max_dim <= 32)
        hipLaunchKernelGGL((k_mcp<T, 32, TS>), dim3(nblocks), dim3(MCP_NT), 0, st, ref, rstride, dst, dstride, blocks, chroma, out_raw);
    else
        hipLaunchKernelGGL((k_mcp<T, 64, TS>), dim3(nblocks), dim3(MCP_NT), 0, st, ref, rstride, dst, dstride, blocks, chroma, out_raw);
}

static int check_blocks_args(SvtAmdContext *ctx, const void *a, const void *b, const void *c, uint32_t n, int bps)
{
    if (!ctx || !a || !b || !c || !n || (bps != 1 && bps != 2))
        return SVT_AMD_ERR_BAD_PARAM;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_mcp_batch_sized(SvtAmdContext *ctx, int bytes_per_sample, int chroma, int out_raw, const void *d_ref,
                                       uint32_t refStride, void *d_dst, uint32_t dstStride, const SvtAmdMcpBlock *d_blocks,
                                       uint32_t nblocks, uint32_t max_block_dim)
{
    int rc = check_blocks_args(ctx, d_ref, d_dst, d_blocks, nblocks, bytes_per_sample);
    if (rc || max_block_dim > 64)
        return rc ? rc : SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    if (bytes_per_sample == 1)
        launch_mcp<uint8_t>(ctx->stream, nblocks, max_block_dim, (const uint8_t *)d_ref, (int)refStride, d_dst, (int)dstStride,
                            (const McpBlock *)d_blocks, chroma, out_raw);
    else
        launch_mcp<uint16_t>(ctx->stream, nblocks, max_block_dim, (const uint16_t *)d_ref, (int)refStride, d_dst, (int)dstStride,
                             (const McpBlock *)d_blocks, chroma, out_raw);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
extern "C" int svt_amd_mcp_batch(SvtAmdContext *ctx, int bytes_per_sample, int chroma, int out_raw, const void *d_ref,
                                 uint32_t refStride, void *d_dst, uint32_t dstStride, const SvtAmdMcpBlock *d_blocks,
                                 uint32_t nblocks)
{
    return svt_amd_mcp_batch_sized(ctx, bytes_per_sample, chroma, out_raw, d_ref, refStride, d_dst, dstStride, d_blocks, nblocks, 0);
}

extern "C" int svt_amd_bipred_clip_batch(SvtAmdContext *ctx, int bytes_per_sample, const int16_t *d_l0, const int16_t *d_l1,
                                         void *d_dst, uint32_t dstStride, int32_t offset, const SvtAmdBiPredBlock *d_blocks,
                                         uint32_t nblocks)
{
    int rc = check_blocks_args(ctx, d_l0, d_l1, d_blocks, nblocks, bytes_per_sample);
    if (rc || !d_dst)
        return rc ? rc : SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    if (bytes_per_sample == 1)
        hipLaunchKernelGGL(k_bipred_clip<uint8_t>, dim3(nblocks), dim3(256), 0, ctx->stream, d_l0, d_l1, (uint8_t *)d_dst,
                           (int)dstStride, (const BiBlock *)d_blocks, offset);
    else
        hipLaunchKernelGGL(k_bipred_clip<uint16_t>, dim3(nblocks), dim3(256), 0, ctx->stream, d_l0, d_l1, (uint16_t *)d_dst,
                           (int)dstStride, (const BiBlock *)d_blocks, offset);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* ---------------- LEAF wrappers (host pointers, reference signatures) ---------------- */
/* exact extent of reference samples the (fx,fy) position reads, relative to the block */
static void tap_extent(int chroma, int f, int *lo, int *hi)
{
    if (!f) {
        *lo = 0, *hi = 0;
    } else if (chroma) {
        *lo = -1, *hi = 2;
    } else {
        *lo = f == 3 ? -2 : -3, *hi = f == 1 ? 3 : 4;
    }
}

template <typename T>
static void mcp_leaf(int chroma, int out_raw, int fx, int fy, const T *refPic, uint32_t srcStride, void *dst, uint32_t dstStride,
                     uint32_t w, uint32_t h)
{
    if (!w || !h || w > 64 || h > 64) {
        svt_amd_set_error("mcp leaf: block %ux%u not supported", w, h);
        return;
    }
    int xlo, xhi, ylo, yhi;
    tap_extent(chroma, fx, &xlo, &xhi);
    tap_extent(chroma, fy, &ylo, &yhi);
    const T *first = refPic + (ptrdiff_t)ylo * srcStride + xlo;
    const size_t rsamples = span(srcStride, w + xhi - xlo, h + yhi - ylo);
    const size_t dbytes = out_raw ? (size_t)w * h * 2 : span(dstStride, w, h) * sizeof(T);
    DBuf r(first, rsamples * sizeof(T)), d(dst, dbytes), bl(nullptr, sizeof(McpBlock), false);
    if (!(r.ok && d.ok && bl.ok))
        return;
    McpBlock hb = {(int32_t)(-(ptrdiff_t)ylo * srcStride - xlo), 0, (uint16_t)w, (uint16_t)h, (uint8_t)fx, (uint8_t)fy, {0, 0}};
    if (hipMemcpy(bl.d, &hb, sizeof(hb), hipMemcpyHostToDevice) != hipSuccess)
        return;
    hipLaunchKernelGGL((k_mcp<T, 64>), dim3(1), dim3(MCP_NT), 0, 0, (const T *)r.d, (int)srcStride, (void *)d.d, (int)dstStride,
                       (const McpBlock *)bl.d, chroma, out_raw);
    if (finish("mcp"))
        d.download(dst, dbytes);
}

#define LUMA_POS(M) M(a, 1, 0) M(b, 2, 0) M(c, 3, 0) M(d, 0, 1) M(e, 1, 1) M(f, 2, 1) M(g, 3, 1) M(h, 0, 2) M(i, 1, 2) \
    M(j, 2, 2) M(k, 3, 2) M(n, 0, 3) M(p, 1, 3) M(q, 2, 3) M(r, 3, 3)

#define UNI8(pos, fx, fy)                                                                                                   \
    extern "C" void svt_amd_LumaInterpolationFilterPos##pos##New(uint8_t *refPic, uint32_t srcStride, uint8_t *dst,         \
                                                                 uint32_t dstStride, uint32_t puWidth, uint32_t puHeight,   \
                                                                 int16_t *firstPassIFDst)                                   \
    { (void)firstPassIFDst; mcp_leaf<uint8_t>(0, 0, fx, fy, refPic, srcStride, dst, dstStride, puWidth, puHeight); }
#define UNI16(pos, fx, fy)                                                                                                  \
    extern "C" void svt_amd_LumaInterpolationFilterPos##pos##New16bit(uint16_t *refPic, uint32_t srcStride, uint16_t *dst,  \
                                                                      uint32_t dstStride, uint32_t puWidth,                 \
                                                                      uint32_t puHeight, int16_t *firstPassIFDst)           \
    { (void)firstPassIFDst; mcp_leaf<uint16_t>(0, 0, fx, fy, refPic, srcStride, dst, dstStride, puWidth, puHeight); }
#define RAW8(pos, fx, fy)                                                                                                   \
    extern "C" void svt_amd_LumaInterpolationFilterPos##pos##OutRaw(uint8_t *refPic, uint32_t srcStride, int16_t *dst,      \
                                                                    uint32_t puWidth, uint32_t puHeight,                    \
                                                                    int16_t *firstPassIFDst)                                \
    { (void)firstPassIFDst; mcp_leaf<uint8_t>(0, 1, fx, fy, refPic, srcStride, dst, 0, puWidth, puHeight); }
#define RAW16(pos, fx, fy)                                                                                                  \
    extern "C" void svt_amd_LumaInterpolationFilterPos##pos##OutRaw16bit(uint16_t *refPic, uint32_t srcStride, int16_t *dst, \
                                                                         uint32_t puWidth, uint32_t puHeight,               \
                                                                         int16_t *firstPassIFDst)                           \
    { (void)firstPassIFDst; mcp_leaf<uint16_t>(0, 1, fx, fy, refPic, srcStride, dst, 0, puWidth, puHeight); }
LUMA_POS(UNI8)
LUMA_POS(UNI16)
LUMA_POS(RAW8)
LUMA_POS(RAW16)

extern "C" void svt_amd_LumaInterpolationCopy(uint8_t *refPic, uint32_t srcStride, uint8_t *dst, uint32_t dstStride,
                                              uint32_t puWidth, uint32_t puHeight, int16_t *firstPassIFDst)
{ (void)firstPassIFDst; mcp_leaf<uint8_t>(0, 0, 0, 0, refPic, srcStride, dst, dstStride, puWidth, puHeight); }
extern "C" void svt_amd_LumaInterpolationCopy16bit(uint16_t *refPic, uint32_t srcStride, uint16_t *dst, uint32_t dstStride,
                                                   uint32_t puWidth, uint32_t puHeight, int16_t *firstPassIFDst)
{ (void)firstPassIFDst; mcp_leaf<uint16_t>(0, 0, 0, 0, refPic, srcStride, dst, dstStride, puWidth, puHeight); }
extern "C" void svt_amd_LumaInterpolationCopyOutRaw(uint8_t *refPic, uint32_t srcStride, int16_t *dst, uint32_t puWidth,
                                                    uint32_t puHeight, int16_t *firstPassIFDst)
{ (void)firstPassIFDst; mcp_leaf<uint8_t>(0, 1, 0, 0, refPic, srcStride, dst, 0, puWidth, puHeight); }
extern "C" void svt_amd_LumaInterpolationCopyOutRaw16bit(uint16_t *refPic, uint32_t srcStride, int16_t *dst, uint32_t puWidth,
                                                         uint32_t puHeight, int16_t *firstPassIFDst)
{ (void)firstPassIFDst; mcp_leaf<uint16_t>(0, 1, 0, 0, refPic, srcStride, dst, 0, puWidth, puHeight); }

/* chroma: Copy ignores the fractions, OneD filters along x when fracPosx != 0 else along y, TwoD both */
#define CH_UNI(name, T, FX, FY)                                                                                             \
    extern "C" void svt_amd_##name(T *refPic, uint32_t srcStride, T *dst, uint32_t dstStride, uint32_t puWidth,             \
                                   uint32_t puHeight, int16_t *firstPassIFDst, uint32_t fracPosx, uint32_t fracPosy)        \
    { (void)firstPassIFDst; (void)fracPosx; (void)fracPosy;                                                                 \
      mcp_leaf<T>(1, 0, (int)(FX), (int)(FY), refPic, srcStride, dst, dstStride, puWidth, puHeight); }
#define CH_RAW(name, T, FX, FY)                                                                                             \
    extern "C" void svt_amd_##name(T *refPic, uint32_t srcStride, int16_t *dst, uint32_t puWidth, uint32_t puHeight,        \
                                   int16_t *firstPassIFDst, uint32_t fracPosx, uint32_t fracPosy)                           \
    { (void)firstPassIFDst; (void)fracPosx; (void)fracPosy;                                                                 \
      mcp_leaf<T>(1, 1, (int)(FX), (int)(FY), refPic, srcStride, dst, 0, puWidth, puHeight); }
CH_UNI(ChromaInterpolationCopy, uint8_t, 0, 0)
CH_UNI(ChromaInterpolationFilterOneD, uint8_t, fracPosx & 7, fracPosx ? 0 : (fracPosy & 7))
CH_UNI(ChromaInterpolationFilterTwoD, uint8_t, fracPosx & 7, fracPosy & 7)
CH_UNI(ChromaInterpolationCopy16bit, uint16_t, 0, 0)
CH_UNI(ChromaInterpolationFilterOneD16bit, uint16_t, fracPosx & 7, fracPosx ? 0 : (fracPosy & 7))
CH_UNI(ChromaInterpolationFilterTwoD16bit, uint16_t, fracPosx & 7, fracPosy & 7)
CH_RAW(ChromaInterpolationCopyOutRaw, uint8_t, 0, 0)
CH_RAW(ChromaInterpolationFilterOneDOutRaw, uint8_t, fracPosx & 7, fracPosx ? 0 : (fracPosy & 7))
CH_RAW(ChromaInterpolationFilterTwoDOutRaw, uint8_t, fracPosx & 7, fracPosy & 7)
CH_RAW(ChromaInterpolationCopyOutRaw16bit, uint16_t, 0, 0)
CH_RAW(ChromaInterpolationFilterOneDOutRaw16bit, uint16_t, fracPosx & 7, fracPosx ? 0 : (fracPosy & 7))
CH_RAW(ChromaInterpolationFilterTwoDOutRaw16bit, uint16_t, fracPosx & 7, fracPosy & 7)

template <typename T>
static void bipred_leaf(uint32_t w, uint32_t h, int16_t *l0, int16_t *l1, T *dst, uint32_t dstStride, int32_t offset)
{
    const size_t n = (size_t)w * h, dbytes = span(dstStride, w, h) * sizeof(T);
    DBuf a(l0, n * 2), b(l1, n * 2), d(dst, dbytes), bl(nullptr, sizeof(BiBlock), false);
    if (!(a.ok && b.ok && d.ok && bl.ok) || !n)
        return;
    BiBlock hb = {0, 0, 0, (uint16_t)w, (uint16_t)h};
    if (hipMemcpy(bl.d, &hb, sizeof(hb), hipMemcpyHostToDevice) != hipSuccess)
        return;
    hipLaunchKernelGGL(k_bipred_clip<T>, dim3(1), dim3(256), 0, 0, (const int16_t *)a.d, (const int16_t *)b.d, (T *)d.d,
                       (int)dstStride, (const BiBlock *)bl.d, offset);
    if (finish("BiPredClipping"))
        d.download(dst, dbytes);
}
extern "C" void svt_amd_BiPredClipping(uint32_t puWidth, uint32_t puHeight, int16_t *list0Src, int16_t *list1Src,
                                       uint8_t *dst, uint32_t dstStride, int32_t offset)
{ bipred_leaf<uint8_t>(puWidth, puHeight, list0Src, list1Src, dst, dstStride, offset); }
extern "C" void svt_amd_BiPredClipping16bit(uint32_t puWidth, uint32_t puHeight, int16_t *list0Src, int16_t *list1Src,
                                            uint16_t *dst, uint32_t dstStride)
{ bipred_leaf<uint16_t>(puWidth, puHeight, list0Src, list1Src, dst, dstStride, 0); }

/* ------------------------------------------------------------------------- */
/* Encode-pass inter prediction of prediction units (driver over the kernels above) */
/* ------------------------------------------------------------------------- */
/* EncodePassInterPrediction (Codec/EbInterPrediction.c:761-926) with EncodeUniPredInterpolation /
 * EncodeBiPredInterpolation (Codec/EbMcp.c:175-250, :562-760), 8-bit 4:2:0.  The host side only turns every unit into
 * interpolation blocks (position clamp :802-812, integer / fractional split, chroma derivation) - a few integer
 * operations per unit; all sample work runs in k_mcp / k_bipred_clip on lists of blocks: per reference list and plane one
 * launch for the uni-predicted units (straight into the prediction planes), one for the 14-bit intermediates of the
 * bi-predicted ones, then one averaging launch per plane. */
#include <vector>

static inline int clampi(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }

template <typename T, typename TS = T>
static int inter_pu_batch(SvtAmdContext *ctx, const SvtAmdInterPuJob *jobs, uint32_t njobs, const SvtAmdRefPicture *ref0,
                          const SvtAmdRefPicture *ref1, T *d_pred_y, uint32_t strideY, T *d_pred_cb, T *d_pred_cr, uint32_t strideC)
{
    if (!ctx || !jobs || !njobs || !d_pred_y || !d_pred_cb || !d_pred_cr || (!ref0 && !ref1))
        return SVT_AMD_ERR_BAD_PARAM;
    const SvtAmdRefPicture *refs[2] = {ref0, ref1};
    std::vector<McpBlock> uni[2][3], raw[2][3];
    std::vector<BiBlock> bi[3];
    size_t raw_len[3] = {0, 0, 0}; /* int16 per list and plane */
    for (uint32_t i = 0; i < njobs; i++) {
        const SvtAmdInterPuJob &J = jobs[i];
        if (J.pred_dir > 2 || J.pu_w < 8 || J.pu_h < 8 || J.pu_w > 64 || J.pu_h > 64 || (J.pu_w & 1) || (J.pu_h & 1))
            return SVT_AMD_ERR_BAD_PARAM;
        const bool is_bi = J.pred_dir == 2;
        for (int l = 0; l < 2; l++) {
            if (!(is_bi || J.pred_dir == l))
                continue;
            const SvtAmdRefPicture *R = refs[l];
            if (!R || !R->d_y || !R->d_cb || !R->d_cr)
                return SVT_AMD_ERR_BAD_PARAM;
            const int px = clampi(((int)R->originX - 71) << 2, (int)(R->width + R->originX + 7) << 2, (((int)J.pu_x + (int)R->originX) << 2) + J.mv[l][0]);
            const int py = clampi(((int)R->originY - 71) << 2, (int)(R->height + R->originY + 7) << 2, (((int)J.pu_y + (int)R->originY) << 2) + J.mv[l][1]);
            for (int p = 0; p < 3; p++) {
                const int sh = p ? 1 : 0;
                McpBlock b;
                b.w = (uint16_t)(J.pu_w >> sh), b.h = (uint16_t)(J.pu_h >> sh), b.pad[0] = b.pad[1] = 0;
                b.fx = (uint8_t)(p ? px & 7 : px & 3), b.fy = (uint8_t)(p ? py & 7 : py & 3);
                b.ref_off = (int32_t)((p ? py >> 3 : py >> 2) * (int)(p ? R->strideC : R->strideY) + (p ? px >> 3 : px >> 2));
                if (is_bi) {
                    b.dst_off = (int32_t)raw_len[p]; /* same offset in both lists' intermediate buffers */
                    raw[l][p].push_back(b);
                } else {
                    b.dst_off = p ? J.dst_off_c : J.dst_off_y;
                    uni[l][p].push_back(b);
                }
            }
        }
        if (is_bi)
            for (int p = 0; p < 3; p++) {
                const int sh = p ? 1 : 0;
                BiBlock c = {(int32_t)raw_len[p], (int32_t)raw_len[p], p ? J.dst_off_c : J.dst_off_y, (uint16_t)(J.pu_w >> sh), (uint16_t)(J.pu_h >> sh)};
                bi[p].push_back(c);
                raw_len[p] += (size_t)c.w * c.h;
            }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    /* one scratch slab: block lists, then the intermediates [list][plane] */
    size_t bytes = 0, off_uni[2][3], off_raw[2][3], off_bi[3], off_int[2][3];
    auto take = [&](size_t n) { const size_t o = bytes; bytes += (n + 255) & ~(size_t)255; return o; };
    for (int l = 0; l < 2; l++)
        for (int p = 0; p < 3; p++)
            off_uni[l][p] = take(uni[l][p].size() * sizeof(McpBlock)), off_raw[l][p] = take(raw[l][p].size() * sizeof(McpBlock));
    for (int p = 0; p < 3; p++)
        off_bi[p] = take(bi[p].size() * sizeof(BiBlock));
    for (int l = 0; l < 2; l++)
        for (int p = 0; p < 3; p++)
            off_int[l][p] = take(raw_len[p] * sizeof(int16_t));
    uint8_t *d_slab = nullptr; /* the context's own scratch (grow-only; callers serialise per context) */
    {
        int rc_s = svt_amd_ctx_scratch(ctx, bytes, &d_slab);
        if (rc_s)
            return rc_s;
    }
    for (int l = 0; l < 2; l++)
        for (int p = 0; p < 3; p++) {
            if (!uni[l][p].empty())
                HIP_TRY(hipMemcpyAsync(d_slab + off_uni[l][p], uni[l][p].data(), uni[l][p].size() * sizeof(McpBlock), hipMemcpyHostToDevice, ctx->stream));
            if (!raw[l][p].empty())
                HIP_TRY(hipMemcpyAsync(d_slab + off_raw[l][p], raw[l][p].data(), raw[l][p].size() * sizeof(McpBlock), hipMemcpyHostToDevice, ctx->stream));
        }
    for (int p = 0; p < 3; p++)
        if (!bi[p].empty())
            HIP_TRY(hipMemcpyAsync(d_slab + off_bi[p], bi[p].data(), bi[p].size() * sizeof(BiBlock), hipMemcpyHostToDevice, ctx->stream));
    T *dst[3] = {d_pred_y, d_pred_cb, d_pred_cr};
    for (int l = 0; l < 2; l++)
        for (int p = 0; p < 3; p++) {
            const SvtAmdRefPicture *R = refs[l];
            if (!R)
                continue;
            const TS *plane = (const TS *)(p == 0 ? R->d_y : p == 1 ? R->d_cb : R->d_cr);
            const int rs = (int)(p ? R->strideC : R->strideY), ds = (int)(p ? strideC : strideY);
            auto max_dim = [](const std::vector<McpBlock> &v) {
                uint32_t m = 0;
                for (const McpBlock &k : v)
                    m = k.w > m ? k.w : m, m = k.h > m ? k.h : m;
                return m;
            };
            if (!uni[l][p].empty())
                launch_mcp<T, TS>(ctx->stream, (uint32_t)uni[l][p].size(), max_dim(uni[l][p]), plane, rs, (void *)dst[p], ds,
                                    (const McpBlock *)(d_slab + off_uni[l][p]), p != 0, 0);
            if (!raw[l][p].empty())
                launch_mcp<T, TS>(ctx->stream, (uint32_t)raw[l][p].size(), max_dim(raw[l][p]), plane, rs, (void *)(d_slab + off_int[l][p]), 0,
                                    (const McpBlock *)(d_slab + off_raw[l][p]), p != 0, 1);
        }
    for (int p = 0; p < 3; p++)
        if (!bi[p].empty()) /* Offset5 / ChromaOffset5 (Codec/EbDefinitions.h:1022-1030) */
            hipLaunchKernelGGL(k_bipred_clip<T>, dim3((unsigned)bi[p].size()), dim3(256), 0, ctx->stream,
                               (const int16_t *)(d_slab + off_int[0][p]), (const int16_t *)(d_slab + off_int[1][p]), dst[p],
                               (int)(p ? strideC : strideY), (const BiBlock *)(d_slab + off_bi[p]), p ? 64 : 16448);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream)); /* the host vectors back the asynchronous uploads */
    return SVT_AMD_OK;
}
extern "C" int svt_amd_inter_pu_batch(SvtAmdContext *ctx, const SvtAmdInterPuJob *jobs, uint32_t njobs, const SvtAmdRefPicture *ref0,
                                      const SvtAmdRefPicture *ref1, uint8_t *d_pred_y, uint32_t strideY, uint8_t *d_pred_cb,
                                      uint8_t *d_pred_cr, uint32_t strideC)
{
    return inter_pu_batch<uint8_t>(ctx, jobs, njobs, ref0, ref1, d_pred_y, strideY, d_pred_cb, d_pred_cr, strideC);
}
/* EncodePassInterPrediction16bit (Codec/EbInterPrediction.c:928-1110) with UniPredInterpolation16bit / BiPredInterpolation16bit
 * (Codec/EbMcp.c:249, :804): the same driver on 16-bit sample planes (10-bit content) */
extern "C" int svt_amd_inter_pu_batch16bit(SvtAmdContext *ctx, const SvtAmdInterPuJob *jobs, uint32_t njobs, const SvtAmdRefPicture *ref0,
                                           const SvtAmdRefPicture *ref1, uint16_t *d_pred_y, uint32_t strideY, uint16_t *d_pred_cb,
                                           uint16_t *d_pred_cr, uint32_t strideC)
{
    return inter_pu_batch<uint16_t>(ctx, jobs, njobs, ref0, ref1, d_pred_y, strideY, d_pred_cb, d_pred_cr, strideC);
}
/* The mode decision of a 10-bit encode (Inter2Nx2NPuPredictionHevc with is16bit, Codec/EbInterPrediction.c:589-760): 8-bit
 * prediction from the 8 most significant bits of the 16-bit reference pictures (UnPackReferenceBlock :414-457 narrows a
 * (w + 8) x (h + 8) window per unit; here the window is narrowed while it is staged into LDS).  ref0 / ref1: 16-bit planes. */
extern "C" int svt_amd_inter_pu_batch_msb(SvtAmdContext *ctx, const SvtAmdInterPuJob *jobs, uint32_t njobs, const SvtAmdRefPicture *ref0,
                                          const SvtAmdRefPicture *ref1, uint8_t *d_pred_y, uint32_t strideY, uint8_t *d_pred_cb,
                                          uint8_t *d_pred_cr, uint32_t strideC)
{
    return inter_pu_batch<uint8_t, uint16_t>(ctx, jobs, njobs, ref0, ref1, d_pred_y, strideY, d_pred_cb, d_pred_cr, strideC);
}
