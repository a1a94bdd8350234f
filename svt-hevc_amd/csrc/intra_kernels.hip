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
#include <cstring>

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

/* ------------------------------------------------------------------------- */
/* Encode-pass intra prediction of a prediction unit from its neighbours      */
/* ------------------------------------------------------------------------- */
/* GenerateIntraReferenceSamplesEncodePass (Codec/EbIntraPrediction.c:212-757, 16-bit :760) + EncodePassIntraPrediction
 * (:4395, 16-bit :4680), fused: one workgroup per unit.  Phase 1 (one thread per neighbour sample): availability of the
 * 4-sample groups, substitution (every missing sample takes the nearest available one before it in scan order, the
 * leading run the first available one).  Phase 2: [1 2 1] / bilinear smoothing of the luma reference.  Phase 3 (one
 * thread per predicted sample, all three planes): H.265 8.4.4.2 prediction from the chosen reference in LDS.
 * Nothing but the job record is read from HBM and nothing but the prediction is written. */
struct IntraPuJob {                       /* = SvtAmdIntraPuJob */
    uint32_t size;
    uint8_t constrained_intra, strong_smoothing, pic_left, pic_top, pic_right, bottom_left_ok, top_right_ok, luma_mode, chroma_mode,
        mode_tl, mode_left[16], mode_top[16], no_smoothing, pad;
    uint16_t left[3][64], top[3][64], tl[3], pad2;
    int32_t dst_off_y, dst_off_c;
};
#include "intra_device.h"

template <typename T>
__global__ __launch_bounds__(256) void k_intra_pu(const IntraPuJob *__restrict__ jobs, T *__restrict__ pred_y, uint32_t strideY,
                                                  T *__restrict__ pred_cb, T *__restrict__ pred_cr, uint32_t strideC)
{
    constexpr int maxv = sizeof(T) == 1 ? 255 : 1023, mid = sizeof(T) == 1 ? 128 : 512, thr = sizeof(T) == 1 ? 8 : 32;
    __shared__ int16_t border[3][132]; /* scan order: bottom-left ... top-left ... top-right */
    __shared__ int16_t ref[3][132];    /* left top-to-bottom, top-left, top: what the prediction reads */
    __shared__ uint8_t ok[36];         /* availability of the 4-sample groups in scan order */
    __shared__ int s_first, s_dc[3], s_mode[3];
    const IntraPuJob &J = jobs[blockIdx.x];
    const int t = threadIdx.x, N = (int)J.size, nb = N >> 2, lgN = 31 - __clz(N);
    if (t <= 4 * nb) {
        bool a;
        if (t < 2 * nb) { /* left group t covers rows [2N-4-4t, 2N-4t) */
            const int e = J.mode_left[(2 * N - 4 - 4 * t) >> 2];
            a = !(e == 0xFE || (!J.bottom_left_ok && t < nb) || e == 0xFF || J.pic_left || (e == 1 && J.constrained_intra));
        } else if (t == 2 * nb) {
            a = !(J.mode_tl == 0xFF || J.pic_left || J.pic_top || (J.mode_tl == 1 && J.constrained_intra));
        } else {
            const int k = t - 2 * nb - 1, e = J.mode_top[k];
            a = !(e == 0xFE || (!J.top_right_ok && k >= nb) || e == 0xFF || J.pic_top || (J.pic_right && k >= nb) ||
                  (e == 1 && J.constrained_intra));
        }
        ok[t] = a;
    }
    if (t == 0)
        s_first = 1 << 30;
    __syncthreads();
    if (t <= 4 * nb && ok[t])
        atomicMin(&s_first, t);
    __syncthreads();
    const int firstGroup = s_first;
    /* substitution, one thread per (plane, sample in scan order) */
    for (int i = t; i < 3 * 129; i += 256) {
        const int p = i / 129, k = i - p * 129, n = p ? N >> 1 : N, g = p ? 2 : 4;
        if (k > 4 * n)
            continue;
        int v = mid;
        if (firstGroup < (1 << 30)) {
            /* group of sample k in scan order: [0, 2n) left, 2n top-left, (2n, 4n] top */
            auto group_of = [&](int kk) { return kk < 2 * n ? kk / g : kk == 2 * n ? 2 * nb : 2 * nb + 1 + (kk - 2 * n - 1) / g; };
            int src = k;
            while (src >= 0 && !ok[group_of(src)])
                src--;
            if (src < 0) { /* leading run: first sample of the first available group */
                src = firstGroup < 2 * nb ? firstGroup * g : firstGroup == 2 * nb ? 2 * n : 2 * n + 1 + (firstGroup - 2 * nb - 1) * g;
            }
            v = src < 2 * n ? J.left[p][2 * n - 1 - src] : src == 2 * n ? J.tl[p] : J.top[p][src - 2 * n - 1];
        }
        border[p][k] = (int16_t)v;
    }
    __syncthreads();
    /* reference choice per plane */
    int lmode = J.luma_mode;
    const int cm = J.chroma_mode;
    const int cmode = cm == 0 ? 0 : cm == 1 ? 26 : cm == 2 ? 10 : cm == 3 ? 1 : lmode;
    const int dA = abs(lmode - 10), dB = abs(lmode - 26), dm = dA < dB ? dA : dB;
    const int thrTab = lgN == 2 ? 35 : lgN == 3 ? 7 : lgN == 4 ? 1 : lgN == 5 ? 0 : 10; /* intraLumaFilterTable */
    const bool filt = dm > thrTab && lmode != 1 && !J.no_smoothing;
    const int bl = border[0][0], tlv = border[0][2 * N], tr = border[0][4 * N];
    const bool strong = J.strong_smoothing && N >= 32 && abs(bl + tlv - 2 * border[0][N]) < thr &&
                        abs(tlv + tr - 2 * border[0][3 * N]) < thr;
    for (int i = t; i < 3 * 129; i += 256) {
        const int p = i / 129, k = i - p * 129, n = p ? N >> 1 : N;
        if (k > 4 * n)
            continue;
        int v = border[p][k];
        if (p == 0 && filt) {
            if (strong) {
                if (k > 0 && k < 2 * n)
                    v = ((2 * n - k) * bl + k * tlv + n) >> (lgN + 1);
                else if (k > 2 * n && k < 4 * n)
                    v = ((2 * n - (k - 2 * n)) * tlv + (k - 2 * n) * tr + n) >> (lgN + 1);
            } else if (k > 0 && k < 4 * n) {
                v = (border[0][k - 1] + 2 * v + border[0][k + 1] + 2) >> 2;
            }
        }
        /* scan order -> left top-to-bottom | top-left | top */
        ref[p][k < 2 * n ? 2 * n - 1 - k : k] = (int16_t)v;
    }
    __syncthreads();
    if (t < 3) {
        const int n = t ? N >> 1 : N;
        int dc = 0;
        for (int i = 0; i < n; i++)
            dc += ref[t][i] + ref[t][2 * n + 1 + i];
        s_dc[t] = (dc + n) >> ((t ? lgN - 1 : lgN) + 1);
        s_mode[t] = t ? cmode : lmode;
    }
    __syncthreads();
    const int nY = N * N, nC = nY >> 2;
    for (int i = t; i < nY + 2 * nC; i += 256) {
        const int p = i < nY ? 0 : (i < nY + nC ? 1 : 2), e = p == 0 ? i : (p == 1 ? i - nY : i - nY - nC);
        const int n = p ? N >> 1 : N, lg = p ? lgN - 1 : lgN, y = e >> lg, x = e & (n - 1);
        if (N == 4 && p != 0)
            continue; /* a 4x4 luma partition has no chroma of its own: the pair belongs to the 8x8 coding unit (a size-8 job) */
        const int v = pu_predict(s_mode[p], n, lg, ref[p], x, y, s_dc[p], p == 0, maxv);
        if (p == 0)
            pred_y[J.dst_off_y + (size_t)y * strideY + x] = (T)v;
        else
            (p == 1 ? pred_cb : pred_cr)[J.dst_off_c + (size_t)y * strideC + x] = (T)v;
    }
}

extern "C" int svt_amd_intra_pu_batch(SvtAmdContext *ctx, int bytes_per_sample, const SvtAmdIntraPuJob *d_jobs, uint32_t njobs,
                                      void *d_pred_y, uint32_t strideY, void *d_pred_cb, void *d_pred_cr, uint32_t strideC)
{
    static_assert(sizeof(IntraPuJob) == sizeof(SvtAmdIntraPuJob), "job layout");
    if (!ctx || !d_jobs || !njobs || !d_pred_y || !d_pred_cb || !d_pred_cr || (bytes_per_sample != 1 && bytes_per_sample != 2))
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    if (bytes_per_sample == 1)
        hipLaunchKernelGGL(k_intra_pu<uint8_t>, dim3(njobs), dim3(256), 0, ctx->stream, (const IntraPuJob *)d_jobs, (uint8_t *)d_pred_y,
                           strideY, (uint8_t *)d_pred_cb, (uint8_t *)d_pred_cr, strideC);
    else
        hipLaunchKernelGGL(k_intra_pu<uint16_t>, dim3(njobs), dim3(256), 0, ctx->stream, (const IntraPuJob *)d_jobs, (uint16_t *)d_pred_y,
                           strideY, (uint16_t *)d_pred_cb, (uint16_t *)d_pred_cr, strideC);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* host-pointer form for one unit: job up, the three blocks back (the binding of the two table slots); blocking */
extern "C" int svt_amd_intra_pu(SvtAmdContext *ctx, int bytes_per_sample, const SvtAmdIntraPuJob *job, void *pred_y, uint32_t strideY,
                                void *pred_cb, void *pred_cr, uint32_t strideC)
{
    const bool wantY = pred_y != nullptr, wantC = pred_cb != nullptr && pred_cr != nullptr;
    if (!ctx || !job || (!wantY && !wantC) || (!pred_cb) != (!pred_cr) || (bytes_per_sample != 1 && bytes_per_sample != 2) ||
        (job->size != 4 && job->size != 8 && job->size != 16 && job->size != 32) || (job->size == 4 && wantC) ||
        (wantY && strideY < job->size) || (wantC && strideC < job->size / 2))
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *d_scratch = nullptr; /* job | Y 32x32 | Cb 16x16 | Cr 16x16 (16-bit worst case) */
    const size_t o_job = 0, o_y = 1024, o_cb = o_y + 2048, o_cr = o_cb + 512, total = o_cr + 512;
    {
        int rc_s = svt_amd_ctx_scratch(ctx, total, &d_scratch);
        if (rc_s)
            return rc_s;
    }
    SvtAmdIntraPuJob j = *job;
    j.dst_off_y = 0, j.dst_off_c = 0;
    const uint32_t N = job->size, C = N / 2;
    const size_t bps = (size_t)bytes_per_sample;
    HIP_TRY(hipMemcpyAsync(d_scratch + o_job, &j, sizeof(j), hipMemcpyHostToDevice, ctx->stream));
    int rc = svt_amd_intra_pu_batch(ctx, bytes_per_sample, (const SvtAmdIntraPuJob *)(d_scratch + o_job), 1, d_scratch + o_y, N,
                                    d_scratch + o_cb, d_scratch + o_cr, C);
    if (rc)
        return rc;
    uint8_t hy[2048], hcb[512], hcr[512];
    if (wantY)
        HIP_TRY(hipMemcpyAsync(hy, d_scratch + o_y, (size_t)N * N * bps, hipMemcpyDeviceToHost, ctx->stream));
    if (wantC) {
        HIP_TRY(hipMemcpyAsync(hcb, d_scratch + o_cb, (size_t)C * C * bps, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(hcr, d_scratch + o_cr, (size_t)C * C * bps, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (uint32_t y = 0; wantY && y < N; y++)
        ::memcpy((uint8_t *)pred_y + (size_t)y * strideY * bps, hy + (size_t)y * N * bps, (size_t)N * bps);
    for (uint32_t y = 0; wantC && y < C; y++) {
        ::memcpy((uint8_t *)pred_cb + (size_t)y * strideC * bps, hcb + (size_t)y * C * bps, (size_t)C * bps);
        ::memcpy((uint8_t *)pred_cr + (size_t)y * strideC * bps, hcr + (size_t)y * C * bps, (size_t)C * bps);
    }
    return SVT_AMD_OK;
}
