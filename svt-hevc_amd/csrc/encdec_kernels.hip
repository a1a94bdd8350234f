/*
 * Device-resident encode pass: the batched boundary `hip_encdec_segment` of SURVEY.md 8(b) for the final encode pass.
 *
 * Replaces, per LCU, the coding-unit loop of EncodePass (Codec/EbCodingLoop.c:2989, CU loop :3180-4594) for intra coding units:
 *     GenerateIntraReferenceSamplesEncodePass + EncodePassIntraPrediction   (:3292-3352; Codec/EbIntraPrediction.c:212, 4395)
 *     EncodeLoop      PictureResidual -> EstimateTransform -> UnifiedQuantizeInvQuantize, Y / Cb / Cr   (:651-1080)
 *     EncodeGenerateRecon   EncodeInvTransform (DC-only shortcut) + PictureAdditionKernel               (:1084-1243)
 *     EncodePassUpdateReconSampleNeighborArrays / ...IntraModeNeighborArrays                              (:3437-3520)
 * ONE launch serves every LCU the host declares ready (the LCUs of one wavefront step of AssignEncDecSegments,
 * Codec/EbEncDecProcess.c:1540): one 256-thread workgroup per LCU walks the LCU's final coding-unit list in order - the closed loop
 * inside an LCU (a unit predicts from the reconstruction of the units before it) runs on the device, nothing returns to the host
 * between units.
 *
 * Device-resident state of a picture (SvtAmdEncDecPicture):
 *   - the reconstruction BEFORE deblocking, three planes: the reference keeps only the last row / column of every unit in its
 *     "neighbour arrays" (epLumaReconNeighborArray ..., Codec/EbNeighborArrays.c:113) because its picture buffer is deblocked in
 *     place; a unit's left / top / top-left neighbours are always on the bottom row or right column of the unit that holds them and
 *     that unit is the last writer of the array entry (Z-order is monotone in x and y), so reading the un-deblocked picture at
 *     (x0 - 1, y), (x, y0 - 1), (x0 - 1, y0 - 1) returns exactly the array contents wherever the reference may read them;
 *   - the mode-type map, one byte per 4x4 luma block (0xFF not coded yet, 1 INTER, 2 INTRA) = epModeTypeNeighborArray with the same
 *     last-writer argument; it decides the availability of every 4-sample neighbour group (constrained intra included).
 * Per LCU the host sends the coding-unit list + the source samples (SvtAmdLcuWork) and gets back the quantised coefficients in the
 * layout of LargestCodingUnit_t.quantizedCoeff, the cbf / DC-only / count fields of every TransformUnit_t and the un-deblocked
 * reconstruction of the LCU (SvtAmdLcuResult): the EncDec output contract of SURVEY 8(a) for the encode pass.
 *
 * Transform unit on N lanes of one wave (Y on wave 0, Cb on wave 1, Cr on wave 2, concurrently): row r of source and prediction ->
 * residual -> forward "Estimate" DCT in registers -> column r quantised / de-quantised in registers -> inverse DCT -> + prediction.
 */
#include "txfm_device.h"
#include "intra_device.h"

struct EpPicture {                 /* = SvtAmdEncDecPicture's device part */
    uint8_t *rec[3];               /* un-deblocked reconstruction, sample (0,0); bytes_per_sample bytes per sample */
    uint32_t pitch[3];             /* samples */
    uint8_t *mode_map;             /* (height / 4) rows of map_pitch bytes */
    uint32_t map_pitch;
    uint16_t width, height;        /* luma */
    uint32_t bps;
};
struct SvtAmdEncDecPicture {
    EpPicture d;
    size_t plane_bytes[3], map_bytes;
    int device;
};

typedef SvtAmdLcuCu LcuCu;
typedef SvtAmdLcuWork LcuWork;
typedef SvtAmdLcuResult LcuResult;

__device__ __forceinline__ int ep_mode_at(const EpPicture &P, int px, int py)
{
    if (px < 0 || py < 0 || px >= (int)P.width || py >= (int)P.height)
        return 0xFE; /* beyond the neighbour array */
    return P.mode_map[(size_t)(py >> 2) * P.map_pitch + (px >> 2)];
}

/* The intra reference of the unit (availability, substitution, smoothing) and the three predicted blocks, written into the
 * reconstruction planes at the unit's position - k_intra_pu (intra_kernels.hip) with the neighbours read from the picture. */
template <typename T>
__device__ void ep_intra_predict(const EpPicture &P, const LcuWork &W, const LcuCu &cu, int t, int16_t (*border)[132], int16_t (*ref)[132],
                                 uint8_t *ok, int *s_small /* [0] first group, [1..3] dc, [4..6] mode */)
{
    constexpr int maxv = sizeof(T) == 1 ? 255 : 1023, mid = sizeof(T) == 1 ? 128 : 512, thr = sizeof(T) == 1 ? 8 : 32;
    const int N = cu.size, nb = N >> 2, lgN = 31 - __clz(N);
    const int x0 = W.lcu_x + cu.x, y0 = W.lcu_y + cu.y;
    const bool pic_left = W.tile_left && cu.x == 0, pic_top = W.tile_top && cu.y == 0;
    const bool pic_right = W.tile_right && ((cu.x + N) & 63) == 0;
    const T *rp[3] = {(const T *)P.rec[0], (const T *)P.rec[1], (const T *)P.rec[2]};
    if (t <= 4 * nb) {
        bool a;
        if (t < 2 * nb) { /* left group t covers rows [2N-4-4t, 2N-4t) */
            const int e = ep_mode_at(P, x0 - 1, y0 + 2 * N - 4 - 4 * t);
            a = !(e == 0xFE || (!cu.bottom_left_ok && t < nb) || e == 0xFF || pic_left || (e == 1 && W.constrained_intra));
        } else if (t == 2 * nb) {
            const int e = ep_mode_at(P, x0 - 1, y0 - 1);
            a = !(e == 0xFE || e == 0xFF || pic_left || pic_top || (e == 1 && W.constrained_intra));
        } else {
            const int k = t - 2 * nb - 1, e = ep_mode_at(P, x0 + 4 * k, y0 - 1);
            a = !(e == 0xFE || (!cu.top_right_ok && k >= nb) || e == 0xFF || pic_top || (pic_right && k >= nb) ||
                  (e == 1 && W.constrained_intra));
        }
        ok[t] = a;
    }
    if (t == 0)
        s_small[0] = 1 << 30;
    __syncthreads();
    if (t <= 4 * nb && ok[t])
        atomicMin(&s_small[0], t);
    __syncthreads();
    const int firstGroup = s_small[0];
    for (int i = t; i < 3 * 129; i += 256) { /* substitution, one thread per (plane, sample in scan order) */
        const int p = i / 129, k = i - p * 129, n = p ? N >> 1 : N, g = p ? 2 : 4;
        if (k > 4 * n)
            continue;
        int v = mid;
        if (firstGroup < (1 << 30)) {
            auto group_of = [&](int kk) { return kk < 2 * n ? kk / g : kk == 2 * n ? 2 * nb : 2 * nb + 1 + (kk - 2 * n - 1) / g; };
            int src = k;
            while (src >= 0 && !ok[group_of(src)])
                src--;
            if (src < 0)
                src = firstGroup < 2 * nb ? firstGroup * g : firstGroup == 2 * nb ? 2 * n : 2 * n + 1 + (firstGroup - 2 * nb - 1) * g;
            const int xp = p ? x0 >> 1 : x0, yp = p ? y0 >> 1 : y0;
            const size_t pitch = P.pitch[p];
            /* scan order: [0, 2n) = left column bottom to top (sample 2n-1-src from the top), 2n = top-left, then the top row */
            v = src < 2 * n ? (int)rp[p][(size_t)(yp + 2 * n - 1 - src) * pitch + xp - 1]
                : src == 2 * n ? (int)rp[p][(size_t)(yp - 1) * pitch + xp - 1] : (int)rp[p][(size_t)(yp - 1) * pitch + xp + (src - 2 * n - 1)];
        }
        border[p][k] = (int16_t)v;
    }
    __syncthreads();
    const int lmode = cu.intra_luma_mode;
    const int dA = abs(lmode - 10), dB = abs(lmode - 26), dm = dA < dB ? dA : dB;
    const int thrTab = lgN == 2 ? 35 : lgN == 3 ? 7 : lgN == 4 ? 1 : lgN == 5 ? 0 : 10; /* intraLumaFilterTable */
    const bool filt = dm > thrTab && lmode != 1;
    const int bl = border[0][0], tlv = border[0][2 * N], tr = border[0][4 * N];
    const bool strong = W.strong_smoothing && N >= 32 && abs(bl + tlv - 2 * border[0][N]) < thr && abs(tlv + tr - 2 * border[0][3 * N]) < thr;
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
        ref[p][k < 2 * n ? 2 * n - 1 - k : k] = (int16_t)v;
    }
    __syncthreads();
    if (t < 3) {
        const int n = t ? N >> 1 : N;
        int dc = 0;
        for (int i = 0; i < n; i++)
            dc += ref[t][i] + ref[t][2 * n + 1 + i];
        s_small[1 + t] = (dc + n) >> ((t ? lgN - 1 : lgN) + 1);
    }
    __syncthreads();
    T *wp[3] = {(T *)P.rec[0], (T *)P.rec[1], (T *)P.rec[2]};
    const int nY = N * N, nC = nY >> 2;
    for (int i = t; i < nY + 2 * nC; i += 256) {
        const int p = i < nY ? 0 : (i < nY + nC ? 1 : 2), e = p == 0 ? i : (p == 1 ? i - nY : i - nY - nC);
        const int n = p ? N >> 1 : N, lg = p ? lgN - 1 : lgN, y = e >> lg, x = e & (n - 1);
        const int v = pu_predict(lmode /* chroma: EB_INTRA_CHROMA_DM */, n, lg, ref[p], x, y, s_small[1 + p], p == 0, maxv);
        const int xp = p ? x0 >> 1 : x0, yp = p ? y0 >> 1 : y0;
        wp[p][(size_t)(yp + y) * P.pitch[p] + xp + x] = (T)v;
    }
}

/* One transform unit of one plane on lanes r = 0..N-1 of the calling wave (the other lanes idle): EncodeLoop + EncodeGenerateRecon.
 * src: source block (pitch srcPitch); rec: prediction in, reconstruction out; coeff: LargestCodingUnit_t.quantizedCoeff position.
 * Returns (lane 0) nz | only_dc << 16. */
template <int N, typename T>
__device__ __forceinline__ uint32_t ep_encode_unit(int r, bool active, const uint8_t *src, int srcPitch, T *rec, size_t recPitch, int16_t *coeff,
                                                   int coeffPitch, int16_t *tile, int qp, int slice_type, uint32_t dz_offset, bool luma)
{
    constexpr int P = TxRegTile<N>::PITCH;
    constexpr int maxv = sizeof(T) == 1 ? 255 : 1023, LG = N == 32 ? 5 : N == 16 ? 4 : N == 8 ? 3 : 2;
    constexpr int depth = sizeof(T) == 1 ? 8 : 10, inc = sizeof(T) == 1 ? 0 : 2;
    constexpr int fs1 = (N == 32 ? 6 : N == 16 ? 4 : N == 8 ? 2 : 1) + inc, fs2 = N == 4 ? 8 : 9, wrap = N == 32 ? 2 : N == 16 ? 1 : 0;
    constexpr int is1 = 7, is2 = 12 - inc;
    int x[N], pred[N];
    if (active) {
        load_row<N, T>(rec + (size_t)r * recPitch, pred);
#pragma unroll
        for (int j = 0; j < N; j++)
            x[j] = (int)src[r * srcPitch + j] - pred[j];
    } else {
#pragma unroll
        for (int j = 0; j < N; j++)
            x[j] = 0, pred[j] = 0;
    }
    fwd_2d_regs<N>(x, tile, r, fs1, fs2, wrap); /* x[j] = coefficient (j, r) */
    const int qpRem = qp % 6, qpPer = qp / 6;
    const uint32_t QF = qpRem == 0 ? 26214u : qpRem == 1 ? 23302u : qpRem == 2 ? 20560u : qpRem == 3 ? 18396u : qpRem == 4 ? 16384u : 14564u;
    const int FFv = qpRem == 0 ? 40 : qpRem == 1 ? 45 : qpRem == 2 ? 51 : qpRem == 3 ? 57 : qpRem == 4 ? 64 : 72;
    const int tshift = 15 - depth - LG, shiftedQBits = 14 + qpPer + tshift;
    const uint32_t q_offset = ((slice_type == 2 || slice_type == 3) ? 171u : 85u) << (shiftedQBits - 9);
    const uint32_t offs = dz_offset ? (uint32_t)(dz_offset * (1u << shiftedQBits) / 20) : q_offset;
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
        nz += (active && qv != 0);
        if (active)
            coeff[j * coeffPitch + r] = (int16_t)qv;
    }
#pragma unroll
    for (int o = 1; o < N; o <<= 1)
        nz += __shfl_xor(nz, o);
    const int dc_rec = __shfl(c[0], 0); /* the de-quantised coefficient (0,0): lane 0 holds column 0 */
    /* tuPtr->isOnlyDc (EbCodingLoop.c:792, 879, 1000): one coefficient, at DC, and no 32x32 luma unit */
    const bool only_dc = nz == 1 && dc_rec != 0 && !(luma && N == 32);
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
    if (only_dc) { /* EncodeInvTransform's shortcut (EbTransforms.c:3516-3535): the twice scaled and clipped DC value everywhere */
        int v = clip16i((64 * dc_rec + (1 << (is1 - 1))) >> is1);
        v = clip16i((64 * (int16_t)v + (1 << (is2 - 1))) >> is2);
#pragma unroll
        for (int j = 0; j < N; j++)
            y[j] = v;
    }
    if (active && nz) { /* cbf == 0: the prediction stays (EbCodingLoop.c:1126) */
#pragma unroll
        for (int j = 0; j < N; j++) {
            const int v = pred[j] + y[j];
            y[j] = v < 0 ? 0 : v > maxv ? maxv : v;
        }
        store_row<N, T>(rec + (size_t)r * recPitch, y);
    }
    return nz | ((uint32_t)only_dc << 16);
}

/* lane = lane of the wave; the unit lives on lanes 0..n-1, the rest of the wave are idle virtual units with tiles of their own (the
 * register transform exchanges rows through the unit's LDS tile and every lane takes part in the wave barriers) */
template <typename T>
__device__ __forceinline__ uint32_t ep_encode_plane(int n, int lane, const uint8_t *src, int srcPitch, T *rec, size_t recPitch,
                                                    int16_t *coeff, int coeffPitch, int16_t *tiles, int qp, int slice_type, uint32_t dz, bool luma)
{
    uint32_t o;
    switch (n) {
    case 32: o = ep_encode_unit<32, T>(lane & 31, lane < 32, src, srcPitch, rec, recPitch, coeff, coeffPitch, tiles + (lane >> 5) * TxRegTile<32>::UNIT, qp, slice_type, dz, luma); break;
    case 16: o = ep_encode_unit<16, T>(lane & 15, lane < 16, src, srcPitch, rec, recPitch, coeff, coeffPitch, tiles + (lane >> 4) * TxRegTile<16>::UNIT, qp, slice_type, dz, luma); break;
    case 8: o = ep_encode_unit<8, T>(lane & 7, lane < 8, src, srcPitch, rec, recPitch, coeff, coeffPitch, tiles + (lane >> 3) * TxRegTile<8>::UNIT, qp, slice_type, dz, luma); break;
    default: o = ep_encode_unit<4, T>(lane & 3, lane < 4, src, srcPitch, rec, recPitch, coeff, coeffPitch, tiles + (lane >> 2) * TxRegTile<4>::UNIT, qp, slice_type, dz, luma); break;
    }
    return __shfl(o, 0); /* lane 0 belongs to the live unit */
}

template <typename T>
__global__ __launch_bounds__(256) void k_encode_lcu(EpPicture P, const LcuWork *__restrict__ works, LcuResult *__restrict__ results)
{
    __shared__ int16_t border[3][132], ref[3][132];
    __shared__ uint8_t ok[36];
    __shared__ int s_small[8];
    __shared__ int16_t tiles[3][2 * TxRegTile<32>::UNIT]; /* 64 / N units of TxRegTile<N>::UNIT each fit for every N */
    const LcuWork &W = works[blockIdx.x];
    LcuResult &R = results[blockIdx.x];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    T *rp[3] = {(T *)P.rec[0], (T *)P.rec[1], (T *)P.rec[2]};
    for (int ci = 0; ci < W.num_cus; ci++) {
        const LcuCu cu = W.cu[ci];
        const int N = cu.size, x0 = W.lcu_x + cu.x, y0 = W.lcu_y + cu.y;
        if (cu.pred_mode == 2 && N <= 32) {
            ep_intra_predict<T>(P, W, cu, t, border, ref, ok, s_small);
            __syncthreads(); /* the prediction is in the picture; the unit's lanes read it back row-wise */
            if (wave < 3) {
                const int p = wave, n = p ? N >> 1 : N;
                const int xp = p ? x0 >> 1 : x0, yp = p ? y0 >> 1 : y0, lx = p ? cu.x >> 1 : cu.x, ly = p ? cu.y >> 1 : cu.y;
                const uint8_t *src = p == 0 ? W.src_y + ly * 64 + lx : (p == 1 ? W.src_cb : W.src_cr) + ly * 32 + lx;
                int16_t *coeff = p == 0 ? R.coeff_y + ly * 64 + lx : (p == 1 ? R.coeff_cb : R.coeff_cr) + ly * 32 + lx;
                const uint32_t o = ep_encode_plane<T>(n, lane, src, p ? 32 : 64, rp[p] + (size_t)yp * P.pitch[p] + xp, P.pitch[p], coeff,
                                                      p ? 32 : 64, tiles[p], p ? cu.chroma_qp : cu.qp, W.slice_type, p ? 0u : cu.dz_offset, p == 0);
                if (lane == 0) {
                    R.cu[ci].nz[p] = (uint16_t)(o & 0xffff);
                    R.cu[ci].cbf[p] = (o & 0xffff) != 0;
                    R.cu[ci].only_dc[p] = (uint8_t)(o >> 16);
                }
            }
        }
        /* EncodePassUpdate...ModeNeighborArrays: the unit is coded now */
        const int cells = N >> 2;
        for (int i = t; i < cells * cells; i += 256)
            P.mode_map[(size_t)((y0 >> 2) + i / cells) * P.map_pitch + (x0 >> 2) + i % cells] = cu.pred_mode;
        __syncthreads(); /* reconstruction and map of this unit are visible to the next one */
    }
    /* the LCU's un-deblocked reconstruction for the host (deblocking / SAO input, reference picture) */
    const int lw = min(64, (int)P.width - (int)W.lcu_x), lh = min(64, (int)P.height - (int)W.lcu_y);
    for (int i = t; i < 64 * 64 + 2 * 32 * 32; i += 256) {
        const int p = i < 4096 ? 0 : (i < 5120 ? 1 : 2), e = p == 0 ? i : (p == 1 ? i - 4096 : i - 5120);
        const int n = p ? 32 : 64, y = e / n, x = e - y * n;
        if (x < (p ? lw >> 1 : lw) && y < (p ? lh >> 1 : lh)) {
            const T v = rp[p][(size_t)((p ? W.lcu_y >> 1 : W.lcu_y) + y) * P.pitch[p] + (p ? W.lcu_x >> 1 : W.lcu_x) + x];
            (p == 0 ? R.rec_y : p == 1 ? R.rec_cb : R.rec_cr)[e] = (uint8_t)v;
        }
    }
}

/* ---- host side ------------------------------------------------------------------------------------------------------------- */
extern "C" int svt_amd_encdec_picture_create(SvtAmdContext *ctx, uint16_t width, uint16_t height, int bytes_per_sample, SvtAmdEncDecPicture **out)
{
    if (!ctx || !out || width < 8 || height < 8 || (width & 7) || (height & 7) || bytes_per_sample != 1) {
        svt_amd_set_error("svt_amd_encdec_picture_create: bad parameter (8-bit pictures, dimensions multiples of 8)");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    SvtAmdEncDecPicture *p = (SvtAmdEncDecPicture *)calloc(1, sizeof(*p));
    if (!p)
        return SVT_AMD_ERR_RESOURCES;
    p->device = ctx->device;
    p->d.width = width, p->d.height = height, p->d.bps = (uint32_t)bytes_per_sample;
    for (int k = 0; k < 3; k++) {
        const uint32_t w = k ? width >> 1 : width, h = k ? height >> 1 : height;
        p->d.pitch[k] = (w + 127) & ~127u;
        p->plane_bytes[k] = (size_t)p->d.pitch[k] * h * bytes_per_sample;
        if (hipMalloc((void **)&p->d.rec[k], p->plane_bytes[k]) != hipSuccess) {
            svt_amd_set_error("hipMalloc (encode-pass picture) failed");
            return SVT_AMD_ERR_RESOURCES;
        }
    }
    p->d.map_pitch = ((uint32_t)(width >> 2) + 63) & ~63u;
    p->map_bytes = (size_t)p->d.map_pitch * (height >> 2);
    if (hipMalloc((void **)&p->d.mode_map, p->map_bytes) != hipSuccess)
        return SVT_AMD_ERR_RESOURCES;
    *out = p;
    return SVT_AMD_OK;
}

extern "C" int svt_amd_encdec_picture_begin(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic)
{
    if (!ctx || !pic)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(pic->d.mode_map, 0xFF, pic->map_bytes, ctx->stream)); /* nothing coded yet */
    HIP_TRY(hipStreamSynchronize(ctx->stream));                                   /* other lanes may encode the first LCU */
    return SVT_AMD_OK;
}

extern "C" int svt_amd_encdec_picture_destroy(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic)
{
    if (!ctx || !pic)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    (void)hipStreamSynchronize(ctx->stream);
    for (int k = 0; k < 3; k++)
        if (pic->d.rec[k])
            (void)hipFree(pic->d.rec[k]);
    if (pic->d.mode_map)
        (void)hipFree(pic->d.mode_map);
    free(pic);
    return SVT_AMD_OK;
}

/* hip_encdec_segment: works / results are HOST arrays of n LCUs that do not depend on each other (one wavefront step); blocking.
 * Contexts (lanes) may call concurrently for different LCUs of the same picture as long as the wavefront order holds between calls. */
extern "C" int svt_amd_encode_lcus(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works, int n, SvtAmdLcuResult *results)
{
    if (!ctx || !pic || !works || !results || n < 1 || n > 1024)
        return SVT_AMD_ERR_BAD_PARAM;
    for (int i = 0; i < n; i++) {
        if (works[i].num_cus > SVT_AMD_LCU_MAX_CUS || works[i].lcu_x >= pic->d.width || works[i].lcu_y >= pic->d.height || (works[i].lcu_x & 63) ||
            (works[i].lcu_y & 63)) {
            svt_amd_set_error("svt_amd_encode_lcus: bad LCU %d", i);
            return SVT_AMD_ERR_BAD_PARAM;
        }
        for (int c = 0; c < works[i].num_cus; c++) {
            const SvtAmdLcuCu &u = works[i].cu[c];
            if (u.pred_mode != 2 || !(u.size == 8 || u.size == 16 || u.size == 32) || u.intra_luma_mode > 34 || (u.x & (u.size - 1)) || (u.y & (u.size - 1)) ||
                u.x + u.size > 64 || u.y + u.size > 64 || works[i].lcu_x + u.x + u.size > pic->d.width || works[i].lcu_y + u.y + u.size > pic->d.height) {
                svt_amd_set_error("svt_amd_encode_lcus: LCU %d unit %d is not an intra 2Nx2N unit of 8..32 inside the picture", i, c);
                return SVT_AMD_ERR_BAD_PARAM;
            }
        }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *d = nullptr;
    const size_t wb = sizeof(SvtAmdLcuWork) * (size_t)n, rb = sizeof(SvtAmdLcuResult) * (size_t)n, wba = (wb + 255) & ~(size_t)255;
    int rc = svt_amd_ctx_scratch(ctx, wba + rb, &d);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(d, works, wb, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_encode_lcu<uint8_t>, dim3((unsigned)n), dim3(256), 0, ctx->stream, pic->d, (const LcuWork *)d, (LcuResult *)(d + wba));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(results, d + wba, rb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}

/* ---- LCUs encoded by the host: their last row / column and edge mode types enter the device picture --------------------------- */
__global__ __launch_bounds__(256) void k_put_borders(EpPicture P, const SvtAmdLcuBorder *B)
{
    const SvtAmdLcuBorder &b = B[blockIdx.x];
    const int t = threadIdx.x;
    const int lw = min(64, (int)P.width - (int)b.lcu_x), lh = min(64, (int)P.height - (int)b.lcu_y);
    /* 0..63 bottom Y, 64..127 right Y, 128..159 / 160..191 bottom / right Cb, 192..223 / 224..255 Cr */
    const int p = t < 128 ? 0 : (t < 192 ? 1 : 2), e = p == 0 ? t : (p == 1 ? t - 128 : t - 192), n = p ? 32 : 64;
    const bool right = e >= n;
    const int i = right ? e - n : e, w = p ? lw >> 1 : lw, h = p ? lh >> 1 : lh;
    const int x0 = p ? b.lcu_x >> 1 : b.lcu_x, y0 = p ? b.lcu_y >> 1 : b.lcu_y;
    const uint8_t *src = p == 0 ? (right ? b.right_y : b.bottom_y) : p == 1 ? (right ? b.right_cb : b.bottom_cb) : (right ? b.right_cr : b.bottom_cr);
    if (i < (right ? h : w))
        P.rec[p][right ? (size_t)(y0 + i) * P.pitch[p] + x0 + w - 1 : (size_t)(y0 + h - 1) * P.pitch[p] + x0 + i] = src[i];
    if (t < 16 && t < (lw >> 2))
        P.mode_map[(size_t)((b.lcu_y + lh - 1) >> 2) * P.map_pitch + (b.lcu_x >> 2) + t] = b.mode_bottom[t];
    if (t >= 16 && t < 32 && t - 16 < (lh >> 2))
        P.mode_map[(size_t)((b.lcu_y >> 2) + t - 16) * P.map_pitch + ((b.lcu_x + lw - 1) >> 2)] = b.mode_right[t - 16];
}

extern "C" int svt_amd_encdec_picture_put_borders(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuBorder *borders, int n)
{
    if (!ctx || !pic || !borders || n < 1 || n > 4096)
        return SVT_AMD_ERR_BAD_PARAM;
    for (int i = 0; i < n; i++)
        if (borders[i].lcu_x >= pic->d.width || borders[i].lcu_y >= pic->d.height || ((borders[i].lcu_x | borders[i].lcu_y) & 63)) {
            svt_amd_set_error("svt_amd_encdec_picture_put_borders: bad LCU %d", i);
            return SVT_AMD_ERR_BAD_PARAM;
        }
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *d = nullptr;
    int rc = svt_amd_ctx_scratch(ctx, sizeof(SvtAmdLcuBorder) * (size_t)n, &d);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(d, borders, sizeof(SvtAmdLcuBorder) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_put_borders, dim3((unsigned)n), dim3(256), 0, ctx->stream, pic->d, (const SvtAmdLcuBorder *)d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}
