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
#include <string.h>
#include <vector>
#include <mutex>
#include "intra_device.h"
#include "rate_device.h"
#include "pmcore_device.h"

struct EpRefPlanes {               /* one reference picture (SvtAmdRefPicture): device pointers to the START of the padded planes */
    const void *plane[3];
    uint32_t stride[2];            /* luma, chroma; samples */
    int32_t originX, originY, width, height; /* luma */
    int32_t size[2];               /* samples of a luma / chroma plane: the bound of the window loads */
};

struct EpPicture {                 /* = SvtAmdEncDecPicture's device part */
    uint8_t *rec[3];               /* un-deblocked reconstruction, sample (0,0); bytes_per_sample bytes per sample */
    uint32_t pitch[3];             /* samples */
    uint8_t *mode_map;             /* (height / 4) rows of map_pitch bytes */
    uint32_t map_pitch;
    unsigned long long *prof;      /* debug (svt_amd_debug_encdec_profile): 16 shader-clock sums per LCU, or null */
    uint16_t width, height;        /* luma */
    uint32_t bps;
    /* P / B pictures (svt_amd_encdec_picture_set_inter): the reference pictures of list 0 / 1 and the picture's coefficient-rate tables */
    EpRefPlanes ref[2];
    const SvtAmdCabacCost *cost;
};
struct SvtAmdEncDecPicture {
    EpPicture d;
    size_t plane_bytes[3], map_bytes;
    int device;
    unsigned *d_sync; /* [0] ticket counter, [1 + lcu] epoch of the picture-level call that finished the LCU, then the ticket order */
    unsigned epoch;
    int nlcu;
    SvtAmdCabacCost *d_cost;
    bool has_ref[2], has_cost;
    /* the in-loop filters behind the encode pass: the deblocked picture and the picture after SAO live beside the un-deblocked one (the SAO
     * statistics need both, svt_amd_encdec_picture_sao); same pitches as rec[] */
    uint8_t *dbk[3], *fin[3];
    bool deblocked, sao_done;
    /* the finished picture with its padding: a reference picture of later pictures (svt_amd_encdec_picture_reference) */
    uint8_t *refp[3];
    size_t refp_bytes[3];
};

typedef SvtAmdLcuCu LcuCu;
/* the contract structs of a sample type */
template <typename T> struct EpTypes;
template <> struct EpTypes<uint8_t> { typedef SvtAmdLcuWork Work; typedef SvtAmdLcuResult Result; typedef SvtAmdLcuBorder Border; };
template <> struct EpTypes<uint16_t> { typedef SvtAmdLcuWork16 Work; typedef SvtAmdLcuResult16 Result; typedef SvtAmdLcuBorder16 Border; };
static_assert(offsetof(SvtAmdLcuWork, src_y) == offsetof(SvtAmdLcuWork16, src_y) && offsetof(SvtAmdLcuResult, rec_y) == offsetof(SvtAmdLcuResult16, rec_y),
              "the 8- and 16-bit contracts share their heads");

__device__ __forceinline__ int ep_mode_at(const EpPicture &P, int px, int py)
{
    if (px < 0 || py < 0 || px >= (int)P.width || py >= (int)P.height)
        return 0xFE; /* beyond the neighbour array */
    return P.mode_map[(size_t)(py >> 2) * P.map_pitch + (px >> 2)];
}

/* The LCU a workgroup encodes lives in LDS: its three reconstruction planes with a ring of neighbour samples (row -1 from x = -1 to
 * 2n - 1: top-left, top and top-right LCUs; column -1: the left LCU), the mode types of its 4x4 cells with the same ring, and its
 * source samples.  Every per-unit access (neighbour fetch, prediction, residual, reconstruction) is an LDS access; the picture in
 * HBM is read once (ring) and written once (finished LCU) per LCU. */
template <typename T>
struct EpLocal {
    static constexpr int PY = 144, PC = 80, X0 = 16; /* row pitches; column of x = 0 (rows of units start 16-byte aligned) */
    T y[65 * PY];
    T c[2][33 * PC];
    uint8_t mode[3][17 * 36]; /* (cy + 1) * 36 + cx + 1: cy in [-1, 16), cx in [-1, 33]; one copy per plane pipeline (each marks the
                               * units IT has finished: the three waves run apart) */
    T src_y[64 * 64], src_c[2][32 * 32];
    SvtAmdLcuCu cus[SVT_AMD_LCU_MAX_CUS]; /* the unit list: a unit's descriptor is an LDS read, not a trip to HBM in front of every unit */
    __device__ __forceinline__ T *at(int p, int x, int y_) { return p == 0 ? &y[(y_ + 1) * PY + X0 + x] : &c[p - 1][(y_ + 1) * PC + X0 + x]; }
    __device__ __forceinline__ int pitch(int p) const { return p == 0 ? PY : PC; }
    /* mode type at luma sample (x, y) relative to the LCU: what lies below the LCU, right of it (from its first row on) or right of
     * the top-right LCU is never coded before this LCU */
    __device__ __forceinline__ int mode_at(int p, int x, int y_) const
    {
        const int cx = x >> 2, cy = y_ >> 2;
        if (cy >= 16 || cx >= 32 || (cy >= 0 && cx >= 16))
            return 0xFF;
        return mode[p][(cy + 1) * 36 + cx + 1];
    }
};

/* wave-level ordering of LDS traffic: what the lanes of this wave wrote is visible to its other lanes */
#define EP_WAVE_SYNC()                                          \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)

/* The intra reference (availability, substitution, smoothing) and the predicted block of ONE plane of the unit, by ONE wave, written
 * into the local reconstruction plane at the unit's position - k_intra_pu (intra_kernels.hip) with the neighbours read from the
 * LCU in LDS.  A plane only ever reads its own samples and the (input-determined) mode types, so the three planes of an LCU are three
 * independent pipelines over the unit list: no workgroup barrier inside the LCU. */
/* the LCU's flags in registers: read from the work record ONCE (the record is in HBM, a load per unit is a round trip per unit) */
struct EpFlags {
    bool tile_left, tile_top, tile_right, constrained_intra, strong_smoothing;
    int slice_type, lcu_x, lcu_y;
    uint32_t full_lambda, cbf_bits[4];
    bool pm_core;
};

/* scratch of one plane pipeline's motion compensation: a tile of up to 32x32 samples at a time */
template <typename T>
struct EpMcScratch {
    static constexpr int WP = 40;
    T win[39 * WP];        /* reference window: (32 + 7) rows x columns */
    int16_t tmp[39 * 32];  /* horizontally filtered rows */
    int16_t raw[32 * 32];  /* list-0 intermediate of a bi-predicted tile */
};

static __constant__ int8_t c_ep_luma_taps[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1},
                                                   {0, 1, -5, 17, 58, -10, 4, -1}};
static __constant__ int8_t c_ep_chroma_taps[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                                     {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

/* EncodePassInterPrediction[16bit] of ONE plane of a 2Nx2N unit by ONE wave, straight into the LCU's reconstruction plane in LDS
 * (the prediction buffer of EncodePass IS the reconstruction buffer): position clamp (Codec/EbInterPrediction.c:802-812), then per
 * tile of <= 32x32 samples and per reference list the H.265 8.5.3.3.3 separable filter in the reference's fixed-point conventions
 * (mcp_kernels.hip:k_mcp, oracle/svt_oracle_mcp.c).  ONE generic path: the 8- (4-) tap filter in both directions also at integer
 * positions - with the identity taps {64} its two-pass arithmetic reduces exactly to the one-pass and copy forms of the reference
 * ((64 h' + 64 B + 2^(11-s)) >> (12-s) == (h + 32) >> 6 with h' = (h - B 2^s) >> s; raw: 64 h' >> 6 == h').  Samples the
 * reference's own functions never touch (zero taps) are loaded from clamped addresses. */
template <typename T>
__device__ __forceinline__ void ep_inter_predict_plane(const EpPicture &P, EpLocal<T> &L, const EpFlags &F, const LcuCu &cu, int p, int lane, EpMcScratch<T> &M)
{
    constexpr int WP = EpMcScratch<T>::WP;
    constexpr int s1 = sizeof(T) == 1 ? 0 : 2, maxv = sizeof(T) == 1 ? 255 : 1023;
    const bool chroma = p != 0;
    const int B = (sizeof(T) == 2 || !chroma) ? 8192 : 0;
    const int N = cu.size, n = chroma ? N >> 1 : N, tn = n > 32 ? 32 : n, lgt = 31 - __clz(tn);
    const int ntaps = chroma ? 4 : 8, first = chroma ? -1 : -3, rows = tn + ntaps - 1;
    const int lx = chroma ? cu.x >> 1 : cu.x, ly = chroma ? cu.y >> 1 : cu.y;
    const bool bi = cu.inter_dir == 2;
    for (int ty0 = 0; ty0 < n; ty0 += 32)
        for (int tx0 = 0; tx0 < n; tx0 += 32) {
            bool second = false;
            for (int l = 0; l < 2; l++) {
                if (!(bi || cu.inter_dir == l))
                    continue;
                const EpRefPlanes &R = P.ref[l];
                const int qx = min(max(((F.lcu_x + cu.x + R.originX) << 2) + cu.mv[l][0], (R.originX - 71) << 2), (R.width + R.originX + 7) << 2);
                const int qy = min(max(((F.lcu_y + cu.y + R.originY) << 2) + cu.mv[l][1], (R.originY - 71) << 2), (R.height + R.originY + 7) << 2);
                const int ix = (chroma ? qx >> 3 : qx >> 2) + tx0, iy = (chroma ? qy >> 3 : qy >> 2) + ty0;
                const int fx = __builtin_amdgcn_readfirstlane(chroma ? qx & 7 : qx & 3), fy = __builtin_amdgcn_readfirstlane(chroma ? qy & 7 : qy & 3);
                const int stride = (int)R.stride[chroma], last = R.size[chroma] - 1;
                const T *plane = (const T *)R.plane[p];
                int tx[8], tv[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    tx[k] = chroma ? (k < 4 ? (int)c_ep_chroma_taps[fx][k & 3] : 0) : (int)c_ep_luma_taps[fx][k];
                    tv[k] = chroma ? (k < 4 ? (int)c_ep_chroma_taps[fy][k & 3] : 0) : (int)c_ep_luma_taps[fy][k];
                }
                /* window: lane = column; every row's load is issued before the first is stored (one memory latency per window, not one
                 * per row) */
                if (lane < rows) {
                    const int base = (iy + first) * stride + ix + first + lane;
                    T v[39];
#pragma unroll
                    for (int j = 0; j < 39; j++)
                        if (j < rows)
                            v[j] = plane[min(max(base + j * stride, 0), last)];
#pragma unroll
                    for (int j = 0; j < 39; j++)
                        if (j < rows)
                            M.win[j * WP + lane] = v[j];
                }
                EP_WAVE_SYNC();
                /* horizontal pass of every window row: a lane slides the taps over a run of seg outputs (seg + taps - 1 reads) */
                const int seg = tn < 8 ? tn : 8, lgs = tn < 8 ? lgt : 3, spr = tn >> lgs; /* runs per row */
                for (int i = lane; i < rows * spr; i += 64) {
                    const int j = i >> (lgt - lgs), x0 = (i & (spr - 1)) << lgs;
                    int in[15];
#pragma unroll
                    for (int k = 0; k < 15; k++)
                        in[k] = k < seg + ntaps - 1 ? (int)M.win[j * WP + x0 + k] : 0;
#pragma unroll
                    for (int o = 0; o < 8; o++)
                        if (o < seg) {
                            int hs = 0;
#pragma unroll
                            for (int k = 0; k < 8; k++)
                                if (k < ntaps)
                                    hs += tx[k] * in[o + k];
                            M.tmp[j * 32 + x0 + o] = (int16_t)((hs - (B << s1)) >> s1);
                        }
                }
                EP_WAVE_SYNC();
                /* vertical pass: a lane owns a column and a run of rpl rows (rpl + taps - 1 reads) */
                const int rpl = tn >= 8 ? (tn * tn) >> 6 : 1, run = rpl < 1 ? 1 : rpl; /* 32: 16, 16: 4, 8: 1, 4: 1 */
                {
                    const int x = lane & (tn - 1), y0 = (lane >> lgt) * run;
                    if (y0 < tn) {
                        int in[23];
#pragma unroll
                        for (int k = 0; k < 23; k++)
                            in[k] = k < run + ntaps - 1 ? (int)M.tmp[(y0 + k) * 32 + x] : 0;
#pragma unroll
                        for (int o = 0; o < 16; o++)
                            if (o < run) {
                                int sum = 0;
#pragma unroll
                                for (int k = 0; k < 8; k++)
                                    if (k < ntaps)
                                        sum += tv[k] * in[o + k];
                                const int y = y0 + o, i = (y << lgt) + x;
                                T *dst = L.at(p, lx + tx0 + x, ly + ty0 + y);
                                if (!bi) {
                                    *dst = (T)min(maxv, max(0, (sum + (B << 6) + (1 << (11 - s1))) >> (12 - s1)));
                                } else if (!second) {
                                    M.raw[i] = (int16_t)(sum >> 6);
                                } else { /* BiPredClipping / BiPredClipping16bit (Offset5 / ChromaOffset5, Codec/EbDefinitions.h:1022-1030) */
                                    const int a = (int)M.raw[i] + (int)(int16_t)(sum >> 6);
                                    *dst = (T)(sizeof(T) == 1 ? min(255, max(0, (a + (chroma ? 64 : 16448)) >> 7)) : min(1023, max(0, (a + 16400) >> 5)));
                                }
                            }
                    }
                }
                EP_WAVE_SYNC();
                second = true;
            }
        }
}

template <typename T>
__device__ __forceinline__ void ep_intra_predict_plane(EpLocal<T> &L, const EpFlags &W, const LcuCu &cu, int p, int lane, int16_t *border,
                                       int16_t *ref, bool prof, unsigned long long (&ph)[4] /* debug: clocks of 4 sub-phases */)
{
    unsigned long long pc = prof ? __builtin_readcyclecounter() : 0;
#define EP_PH(i)                                                  \
    do {                                                          \
        if (prof) {                                               \
            const unsigned long long now = __builtin_readcyclecounter(); \
            ph[i] += now - pc, pc = now;                          \
        }                                                         \
    } while (0)
    constexpr int maxv = sizeof(T) == 1 ? 255 : 1023, mid = sizeof(T) == 1 ? 128 : 512, thr = sizeof(T) == 1 ? 8 : 32;
    const int N = cu.size, nb = N >> 2, lgN = 31 - __clz(N);
    const int n = p ? N >> 1 : N, lgn = p ? lgN - 1 : lgN, lgG = p ? 1 : 2, g = 1 << lgG; /* plane size; samples per neighbour group */
    const int lx = p ? cu.x >> 1 : cu.x, ly = p ? cu.y >> 1 : cu.y;
    const bool pic_left = W.tile_left && cu.x == 0, pic_top = W.tile_top && cu.y == 0;
    const bool pic_right = W.tile_right && ((cu.x + N) & 63) == 0;
    /* the 4 nb + 1 <= 33 neighbour groups: availability by ballot */
    bool a = false;
    if (lane > 4 * nb) {
    } else if (lane < 2 * nb) { /* left group covers luma rows [2N-4-4 lane, 2N-4 lane) */
        const int e = L.mode_at(p, cu.x - 1, cu.y + 2 * N - 4 - 4 * lane);
        a = !(e == 0xFE || (!cu.bottom_left_ok && lane < nb) || e == 0xFF || pic_left || (e == 1 && W.constrained_intra));
    } else if (lane == 2 * nb) {
        const int e = L.mode_at(p, cu.x - 1, cu.y - 1);
        a = !(e == 0xFE || e == 0xFF || pic_left || pic_top || (e == 1 && W.constrained_intra));
    } else {
        const int k = lane - 2 * nb - 1, e = L.mode_at(p, cu.x + 4 * k, cu.y - 1);
        a = !(e == 0xFE || (!cu.top_right_ok && k >= nb) || e == 0xFF || pic_top || (pic_right && k >= nb) || (e == 1 && W.constrained_intra));
    }
    const unsigned long long m = __ballot(a);
    const int firstGroup = m ? __ffsll((long long)m) - 1 : 1 << 30;
    EP_PH(0);
    for (int k = lane; k <= 4 * n; k += 64) { /* substitution, one lane per sample in scan order */
        int v = mid;
        if (firstGroup < (1 << 30)) {
            /* the nearest available sample at or below k in scan order (the reference walks down sample by sample): k itself when
             * its group is there, otherwise the LAST sample of the nearest available group below - bit operations on the
             * availability mask - and, with nothing below, the first sample of the first available group */
            const int gk = k < 2 * n ? k >> lgG : k == 2 * n ? 2 * nb : 2 * nb + 1 + ((k - 2 * n - 1) >> lgG);
            const unsigned long long below = m & ((2ull << gk) - 1ull); /* groups 0..gk */
            int src;
            if ((below >> gk) & 1ull) {
                src = k;
            } else if (below) {
                const int sg = 63 - __clzll((long long)below);
                src = sg < 2 * nb ? sg * g + g - 1 : sg == 2 * nb ? 2 * n : 2 * n + 1 + (sg - 2 * nb - 1) * g + g - 1;
            } else {
                src = firstGroup < 2 * nb ? firstGroup * g : firstGroup == 2 * nb ? 2 * n : 2 * n + 1 + (firstGroup - 2 * nb - 1) * g;
            }
            /* scan order: [0, 2n) = left column bottom to top (sample 2n-1-src from the top), 2n = top-left, then the top row */
            v = src < 2 * n ? (int)*L.at(p, lx - 1, ly + 2 * n - 1 - src) : src == 2 * n ? (int)*L.at(p, lx - 1, ly - 1)
                                                                                          : (int)*L.at(p, lx + (src - 2 * n - 1), ly - 1);
        }
        border[k] = (int16_t)v;
    }
    EP_WAVE_SYNC();
    EP_PH(1);
    const int lmode = cu.intra_luma_mode;
    const int dA = abs(lmode - 10), dB = abs(lmode - 26), dm = dA < dB ? dA : dB;
    const int thrTab = lgN == 2 ? 35 : lgN == 3 ? 7 : lgN == 4 ? 1 : lgN == 5 ? 0 : 10; /* intraLumaFilterTable */
    const bool filt = p == 0 && dm > thrTab && lmode != 1;
    const int bl = border[0], tlv = border[2 * n], tr = border[4 * n];
    const bool strong = p == 0 && W.strong_smoothing && N >= 32 && abs(bl + tlv - 2 * border[n]) < thr && abs(tlv + tr - 2 * border[3 * n]) < thr;
    for (int k = lane; k <= 4 * n; k += 64) {
        int v = border[k];
        if (filt) {
            if (strong) {
                if (k > 0 && k < 2 * n)
                    v = ((2 * n - k) * bl + k * tlv + n) >> (lgN + 1);
                else if (k > 2 * n && k < 4 * n)
                    v = ((2 * n - (k - 2 * n)) * tlv + (k - 2 * n) * tr + n) >> (lgN + 1);
            } else if (k > 0 && k < 4 * n) {
                v = (border[k - 1] + 2 * v + border[k + 1] + 2) >> 2;
            }
        }
        ref[k < 2 * n ? 2 * n - 1 - k : k] = (int16_t)v;
    }
    EP_WAVE_SYNC();
    EP_PH(2);
    int dcv = 0;
    if (lmode == 1) { /* DC: left column + top row of the plane */
        int dc = lane < n ? ref[lane] + ref[2 * n + 1 + lane] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            dc += __shfl_xor(dc, o);
        dcv = (dc + n) >> (lgn + 1);
    }
    for (int e = lane; e < n * n; e += 64) {
        const int y = e >> lgn, x = e & (n - 1);
        const int v = pu_predict(lmode /* chroma: EB_INTRA_CHROMA_DM */, n, lgn, ref, x, y, dcv, p == 0, maxv);
        *L.at(p, lx + x, ly + y) = (T)v;
    }
    EP_WAVE_SYNC();
    EP_PH(3);
#undef EP_PH
}

/* One transform unit of one plane on lanes r = 0..N-1 of the calling wave (the other lanes idle): EncodeLoop + EncodeGenerateRecon.
 * src: source block (pitch srcPitch); rec: prediction in, reconstruction out; coeff: LargestCodingUnit_t.quantizedCoeff position.
 * Returns (lane 0) nz | only_dc << 16. */
/* the luma cbf decision of an AMVP unit (EbCodingLoop.c:4075-4124): PictureFullDistortionLuma on the coefficients, TuEstimateCoeffBitsEncDec,
 * EncodeTuCalcCost (EbRateDistortionCost.c:2578) */
struct EpDecide {
    const SvtAmdCabacCost *cost;
    int16_t *qbuf;                 /* LDS, N x N: the quantised coefficients of the unit for the rate estimator */
    uint32_t lambda, zero_bits, nonzero_bits; /* fullLambda, lumaCbfBits[ctx], lumaCbfBits[ctx + 5] */
    /* the PM-core quantiser of encMode 1..4 (UnifiedQuantizeInvQuantize with rdoqPmCoreMethod == EB_PMCORE -> DecoupledQuantizeInvQuantizeLoops,
     * Codec/EbTransforms.c:3009-3052, :2605-2973): no dead-zone override; luma levels re-decided per 4x4 block (pmcore_device.h) */
    bool pm_core;
    int cand_type;                 /* predictionModeFlag of the unit */
    int16_t *cfbuf;                /* LDS, N x N: the unit's coefficients for the re-decision */
    int16_t (*Pq)[16];             /* LDS, 64 x 16: its per-lane scratch */
};

/* Returns (every lane) nz | only_dc << 16 | cbf << 17. */
template <int N, typename T>
__device__ __forceinline__ uint32_t ep_encode_unit(int lane, int r, bool active, const T *src, int srcPitch, T *rec, size_t recPitch, int16_t *coeff,
                                                   int coeffPitch, int16_t *tile, int qp, int slice_type, uint32_t dz_offset, bool luma,
                                                   bool decide, const EpDecide &dec)
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
    const uint32_t offs = (dz_offset && !dec.pm_core) ? (uint32_t)(dz_offset * (1u << shiftedQBits) / 20) : q_offset;
    const int shiftedFFunc = qpPer > 8 ? FFv << (qpPer - 2) : FFv << qpPer;
    const int shiftNum = qpPer > 8 ? 20 - 14 - tshift - 2 : 20 - 14 - tshift, iq_offset = 1 << (shiftNum - 1);
    unsigned nz = 0;
    int c[N], q[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        const int v = x[j], sign = v < 0 ? -1 : 1;
        int tq = abs(v);
        tq = (int)((uint32_t)tq * QF);
        tq = (int)((uint32_t)tq + offs);
        tq >>= shiftedQBits;
        q[j] = clip16i(sign * tq);
        nz += (active && q[j] != 0);
    }
#pragma unroll
    for (int o = 1; o < N; o <<= 1)
        nz += __shfl_xor(nz, o);
    if constexpr (N >= 8) {
        if (dec.pm_core && luma) { /* wave-uniform: the 4x4 blocks of the unit on all 64 lanes (a 32x32 unit has 64 of them) */
            if (active) {
#pragma unroll
                for (int j = 0; j < N; j++)
                    dec.cfbuf[j * N + r] = (int16_t)x[j], dec.qbuf[j * N + r] = (int16_t)q[j];
            }
            EP_WAVE_SYNC();
            if (__shfl((int)nz, 0) != 0) {
                FlUnit Q;
                Q.active = 1, Q.base = 0, Q.pitch = N, Q.area = N, Q.lg = LG;
                Q.QF = QF, Q.q_offset = q_offset, Q.shiftedQBits = shiftedQBits, Q.shiftedFFunc = shiftedFFunc, Q.iq_offset = iq_offset, Q.shiftNum = shiftNum;
                pm_core_blocks<64>(*dec.cost, dec.cfbuf, dec.qbuf, N, N, LG, lane, true, lane, dec.cand_type, dec.lambda, Q, dec.Pq);
                EP_WAVE_SYNC();
                nz = 0;
                if (active) {
#pragma unroll
                    for (int j = 0; j < N; j++)
                        q[j] = dec.qbuf[j * N + r], nz += q[j] != 0;
                }
#pragma unroll
                for (int o = 1; o < N; o <<= 1)
                    nz += __shfl_xor(nz, o);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        c[j] = clip16i(((q[j] * shiftedFFunc) + iq_offset) >> shiftNum);
        if (active)
            coeff[j * coeffPitch + r] = (int16_t)q[j];
        if (N >= 8 && decide && active)
            dec.qbuf[j * N + r] = (int16_t)q[j];
    }
    const int dc_rec = __shfl(c[0], 0); /* the de-quantised coefficient (0,0): lane 0 holds column 0 */
    /* tuPtr->isOnlyDc (EbCodingLoop.c:792, 879, 1000): one coefficient, at DC, and no 32x32 luma unit */
    const bool only_dc = nz == 1 && dc_rec != 0 && !(luma && N == 32);
    bool cbf = nz != 0;
    if constexpr (N >= 8) {
        if (decide) { /* wave-uniform */
            /* FullDistortionKernel_32bit / ...CbfZero_32bit (EbPictureOperators_C.c:385-480): 16-bit difference, 32-bit sums; a DC-only
             * unit is measured on its DC alone */
            uint32_t d0 = 0, d1 = 0;
            if (active) {
#pragma unroll
                for (int j = 0; j < N; j++)
                    if (!only_dc || (j == 0 && r == 0)) {
                        const int df = (int16_t)(x[j] - c[j]), cf = (int16_t)x[j];
                        d0 += (uint32_t)(df * df), d1 += (uint32_t)(cf * cf);
                    }
            }
#pragma unroll
            for (int o = 1; o < N; o <<= 1)
                d0 += __shfl_xor(d0, o), d1 += __shfl_xor(d1, o);
            /* the unit's lanes hold the sums; the rate estimator below runs on (N / 4)^2 lanes - all 64 for a 32x32 unit, whose upper
             * half are not lanes of the unit: everything the decision uses comes from lane 0 */
            const uint32_t nzu = (uint32_t)__shfl((int)nz, 0);
            d0 = (uint32_t)__shfl((int)d0, 0), d1 = (uint32_t)__shfl((int)d1, 0);
            constexpr int dshift = 2 * (7 - LG);
            const unsigned long long dz = ((unsigned long long)d1 + (1ull << (dshift - 1))) >> dshift;
            const unsigned long long dn = nzu ? ((unsigned long long)d0 + (1ull << (dshift - 1))) >> dshift : dz;
            EP_WAVE_SYNC(); /* qbuf is written */
            constexpr int S = (N / 4) * (N / 4);
            const SvtAmdTuInfo ti = {nzu, 1 /* INTER_MODE */, 0xFF, 0xFF, 0};
            const uint32_t b32 = coeff_bits_lanes(*dec.cost, dec.qbuf, N, LG, ti, lane < S, lane, lane & (S - 1));
            const unsigned long long tuBits = (((unsigned long long)__shfl(b32, 0)) << 10) >> 15;
            const unsigned long long nzRate = (tuBits << 15) + dec.nonzero_bits, zRate = dec.zero_bits, lam = dec.lambda;
            const unsigned long long zCost = (dz << 8) + (((lam * zRate) + (1u << 22)) >> 23);
            const unsigned long long nzCost = (dn << 8) + (((lam * nzRate) + (1u << 22)) >> 23);
            cbf = nzu != 0 && nzCost < zCost;
        }
    }
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
    if (active && cbf) { /* cbf == 0: the prediction stays (EbCodingLoop.c:1126) */
#pragma unroll
        for (int j = 0; j < N; j++) {
            const int v = pred[j] + y[j];
            y[j] = v < 0 ? 0 : v > maxv ? maxv : v;
        }
        store_row<N, T>(rec + (size_t)r * recPitch, y);
    }
    return nz | ((uint32_t)only_dc << 16) | ((uint32_t)cbf << 17);
}

/* lane = lane of the wave; the unit lives on lanes 0..n-1, the rest of the wave are idle virtual units with tiles of their own (the
 * register transform exchanges rows through the unit's LDS tile and every lane takes part in the wave barriers) */
template <typename T>
__device__ __forceinline__ uint32_t ep_encode_plane(int n, int lane, const T *src, int srcPitch, T *rec, size_t recPitch,
                                                    int16_t *coeff, int coeffPitch, int16_t *tiles, int qp, int slice_type, uint32_t dz, bool luma,
                                                    bool decide = false, const EpDecide &dec = EpDecide())
{
    uint32_t o;
    switch (n) {
    case 32: o = ep_encode_unit<32, T>(lane, lane & 31, lane < 32, src, srcPitch, rec, recPitch, coeff, coeffPitch, tiles + (lane >> 5) * TxRegTile<32>::UNIT, qp, slice_type, dz, luma, decide, dec); break;
    case 16: o = ep_encode_unit<16, T>(lane, lane & 15, lane < 16, src, srcPitch, rec, recPitch, coeff, coeffPitch, tiles + (lane >> 4) * TxRegTile<16>::UNIT, qp, slice_type, dz, luma, decide, dec); break;
    case 8: o = ep_encode_unit<8, T>(lane, lane & 7, lane < 8, src, srcPitch, rec, recPitch, coeff, coeffPitch, tiles + (lane >> 3) * TxRegTile<8>::UNIT, qp, slice_type, dz, luma, decide, dec); break;
    default: o = ep_encode_unit<4, T>(lane, lane & 3, lane < 4, src, srcPitch, rec, recPitch, coeff, coeffPitch, tiles + (lane >> 2) * TxRegTile<4>::UNIT, qp, slice_type, dz, luma, false, dec); break;
    }
    return __shfl(o, 0); /* lane 0 belongs to the live unit */
}

template <typename T>
struct EpShared {
    int16_t border[3][132], ref[3][132];       /* per plane pipeline */
    int16_t tiles[3][2 * TxRegTile<32>::UNIT]; /* 64 / N units of TxRegTile<N>::UNIT each fit for every N */
    EpMcScratch<T> mc[3];                      /* inter units */
    int16_t qbuf[32 * 32];                     /* luma cbf decision of AMVP units; levels of the PM-core re-decision */
    int16_t cfbuf[32 * 32];                    /* PM-core: the luma unit's coefficients */
    int16_t Pq[64][16];                        /* PM-core: per-lane scratch of the 4x4 rate estimate */
};

/* the coding-unit loop of one LCU, by one workgroup of 256 threads */
template <typename T>
__device__ __forceinline__ void ep_encode_lcu(const EpPicture &P, const typename EpTypes<T>::Work &W, typename EpTypes<T>::Result &R, EpShared<T> &S, EpLocal<T> &L)
{
    int16_t (*border)[132] = S.border, (*ref)[132] = S.ref;
    int16_t (*tiles)[2 * TxRegTile<32>::UNIT] = S.tiles;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const T *rp[3] = {(const T *)P.rec[0], (const T *)P.rec[1], (const T *)P.rec[2]};
    T *wp[3] = {(T *)P.rec[0], (T *)P.rec[1], (T *)P.rec[2]};
    unsigned long long c_pred = 0, c_enc = 0, c0 = P.prof ? __builtin_readcyclecounter() : 0, c1 = 0;
    unsigned long long c_ph[4] = {0, 0, 0, 0};
    const int lw = min(64, (int)P.width - (int)W.lcu_x), lh = min(64, (int)P.height - (int)W.lcu_y);
    /* ---- the LCU's surroundings and source into LDS ---- */
    for (int i = t; i < 17 * 36; i += 256) {
        const int cy = i / 36 - 1, cx = i - (cy + 1) * 36 - 1;
        const uint8_t v = (uint8_t)((cy < 0 || cx < 0) ? ep_mode_at(P, (int)W.lcu_x + 4 * cx, (int)W.lcu_y + 4 * cy) : 0xFF);
        L.mode[0][i] = v, L.mode[1][i] = v, L.mode[2][i] = v;
    }
    for (int i = t; i < 130 + 2 * 66 + 64 + 2 * 32; i += 256) { /* ring samples: top rows (x = -1 .. 2n-1), then left columns */
        int p, x, y;
        if (i < 130 + 2 * 66) {
            p = i < 130 ? 0 : (i < 196 ? 1 : 2);
            x = (p == 0 ? i : p == 1 ? i - 130 : i - 196) - 1, y = -1;
        } else {
            const int e = i - (130 + 2 * 66);
            p = e < 64 ? 0 : (e < 96 ? 1 : 2);
            x = -1, y = p == 0 ? e : p == 1 ? e - 64 : e - 96;
        }
        if (x >= (p ? 64 : 128))
            continue;
        const int gx = (p ? W.lcu_x >> 1 : W.lcu_x) + x, gy = (p ? W.lcu_y >> 1 : W.lcu_y) + y;
        const int pw = p ? P.width >> 1 : P.width, ph = p ? P.height >> 1 : P.height;
        if (gx >= 0 && gy >= 0 && gx < pw && gy < ph) /* what is not there is never marked available */
            *L.at(p, x, y) = rp[p][(size_t)gy * P.pitch[p] + gx];
    }
    for (int i = t; i < (64 * 64 + 2 * 32 * 32) * (int)sizeof(T) / 4; i += 256) {
        const uint32_t v = ((const uint32_t *)W.src_y)[i]; /* src_y, src_cb, src_cr are contiguous in the contract */
        ((uint32_t *)L.src_y)[i] = v;
    }
    for (int i = t; i < (int)(sizeof(L.cus) / 4); i += 256)
        ((uint32_t *)L.cus)[i] = ((const uint32_t *)W.cu)[i];
    const int num_cus = W.num_cus;
    const EpFlags F = {W.tile_left != 0, W.tile_top != 0, W.tile_right != 0, W.constrained_intra != 0, W.strong_smoothing != 0, (int)W.slice_type,
                       (int)W.lcu_x,     (int)W.lcu_y,    W.full_lambda,     {W.luma_cbf_bits[0], W.luma_cbf_bits[1], W.luma_cbf_bits[2], W.luma_cbf_bits[3]},
                       W.pm_core != 0};
    __syncthreads();
    if (wave < 3) { /* wave p = plane p: its own pipeline over the unit list (luma is the long one) */
        const int p = wave;
        for (int ci = 0; ci < num_cus; ci++) {
            const LcuCu cu = L.cus[ci];
            const int N = cu.size;
            if (cu.pred_mode == 1) { /* INTER_MODE, 2Nx2N (EbCodingLoop.c:3817-4400) */
                ep_inter_predict_plane<T>(P, L, F, cu, p, lane, S.mc[p]);
                if (P.prof)
                    c1 = __builtin_readcyclecounter(), c_pred += c1 - c0;
                const int ntu = N == 64 ? 4 : 1, TS = N == 64 ? 32 : N, n = p ? TS >> 1 : TS;
                const bool amvp = cu.inter_kind == SVT_AMD_EP_INTER_AMVP;
                const EpDecide D = {P.cost, S.qbuf, F.full_lambda, N == TS ? F.cbf_bits[1] : F.cbf_bits[0], N == TS ? F.cbf_bits[3] : F.cbf_bits[2],
                                    F.pm_core,      1,      S.cfbuf,       S.Pq};
                uint32_t any = 0;
                for (int tu = 0; tu < ntu; tu++) {
                    const int tx = cu.x + ((tu & 1) << 5), ty = cu.y + ((tu >> 1) << 5);
                    const int lx = p ? tx >> 1 : tx, ly = p ? ty >> 1 : ty;
                    uint32_t o = 0;
                    if (cu.inter_kind != SVT_AMD_EP_INTER_SKIP) {
                        const T *src = p == 0 ? L.src_y + ly * 64 + lx : L.src_c[p - 1] + ly * 32 + lx;
                        int16_t *coeff = p == 0 ? R.coeff_y + ly * 64 + lx : (p == 1 ? R.coeff_cb : R.coeff_cr) + ly * 32 + lx;
                        o = ep_encode_plane<T>(n, lane, src, p ? 32 : 64, L.at(p, lx, ly), (size_t)L.pitch(p), coeff, p ? 32 : 64, tiles[p],
                                               (p ? cu.chroma_qp : cu.qp) + (sizeof(T) == 2 ? 12 : 0), F.slice_type, p ? 0u : cu.dz_offset, p == 0,
                                               p == 0 && amvp, D);
                    }
                    if (lane == 0) { /* a 64x64 unit: entries 1..4 = its four transform units */
                        SvtAmdLcuCuResult &E = R.cu[ci + (N == 64 ? 1 + tu : 0)];
                        E.nz[p] = (uint16_t)(o & 0xffff), E.cbf[p] = (uint8_t)((o >> 17) & 1), E.only_dc[p] = (uint8_t)((o >> 16) & 1);
                    }
                    any |= (o >> 17) & 1;
                }
                if (N == 64 && lane == 0) /* transformUnitArray[0]: chroma flags OR-ed (:4263-4281), luma only by EncodeTuCalcCost */
                    R.cu[ci].nz[p] = 0, R.cu[ci].only_dc[p] = 0, R.cu[ci].cbf[p] = (uint8_t)(any && (p != 0 || amvp));
            } else if (cu.pred_mode == 2 && N <= 32) {
                ep_intra_predict_plane<T>(L, F, cu, p, lane, border[p], ref[p], P.prof && p == 0, c_ph);
                if (P.prof)
                    c1 = __builtin_readcyclecounter(), c_pred += c1 - c0;
                const int n = p ? N >> 1 : N;
                const int lx = p ? cu.x >> 1 : cu.x, ly = p ? cu.y >> 1 : cu.y;
                const T *src = p == 0 ? L.src_y + ly * 64 + lx : L.src_c[p - 1] + ly * 32 + lx;
                int16_t *coeff = p == 0 ? R.coeff_y + ly * 64 + lx : (p == 1 ? R.coeff_cb : R.coeff_cr) + ly * 32 + lx;
                const uint32_t o = ep_encode_plane<T>(n, lane, src, p ? 32 : 64, L.at(p, lx, ly), (size_t)L.pitch(p), coeff, p ? 32 : 64, tiles[p],
                                                      (p ? cu.chroma_qp : cu.qp) + (sizeof(T) == 2 ? 12 : 0) /* QP_BD_OFFSET, EbCodingLoop.c:1307 */,
                                                      F.slice_type, p ? 0u : cu.dz_offset, p == 0, false,
                                                      EpDecide{P.cost, S.qbuf, F.full_lambda, 0u, 0u, F.pm_core, 2, S.cfbuf, S.Pq});
                if (lane == 0) {
                    R.cu[ci].nz[p] = (uint16_t)(o & 0xffff);
                    R.cu[ci].cbf[p] = (o & 0xffff) != 0;
                    R.cu[ci].only_dc[p] = (uint8_t)((o >> 16) & 1);
                }
            }
            /* EncodePassUpdate...ModeNeighborArrays: this pipeline has coded the unit */
            const int lgc = 29 - __clz(N), cells = 1 << lgc; /* N / 4 */
            for (int i = lane; i < cells * cells; i += 64)
                L.mode[p][((cu.y >> 2) + (i >> lgc) + 1) * 36 + (cu.x >> 2) + (i & (cells - 1)) + 1] = cu.pred_mode;
            EP_WAVE_SYNC(); /* reconstruction and mode cells of this unit are visible to the next one (same wave) */
            if (P.prof)
                c0 = __builtin_readcyclecounter(), c_enc += c0 - c1;
        }
    }
    __syncthreads(); /* the three planes are done */
    /* ---- the finished LCU leaves LDS: picture planes + mode map (neighbours of later LCUs, the host's deblocking / SAO input and
     * reference picture) and the result record ---- */
    for (int i = t; i < (64 * 64 + 2 * 32 * 32) / 4; i += 256) {
        const int p = i < 1024 ? 0 : (i < 1280 ? 1 : 2), e = p == 0 ? i : (p == 1 ? i - 1024 : i - 1280);
        const int n4 = p ? 8 : 16, y = e / n4, x = (e - y * n4) * 4;
        if (x < (p ? lw >> 1 : lw) && y < (p ? lh >> 1 : lh)) { /* widths are multiples of 8 luma samples: whole groups of 4 */
            const T *q = L.at(p, x, y);
            const int gx = (p ? W.lcu_x >> 1 : W.lcu_x) + x, gy = (p ? W.lcu_y >> 1 : W.lcu_y) + y;
            T *g = wp[p] + (size_t)gy * P.pitch[p] + gx;
            T *r = (p == 0 ? R.rec_y : p == 1 ? R.rec_cb : R.rec_cr) + y * (p ? 32 : 64) + x;
#pragma unroll
            for (int k = 0; k < 4; k++)
                g[k] = q[k], r[k] = q[k];
        }
    }
    for (int i = t; i < 16 * 16; i += 256) {
        const int cy = i >> 4, cx = i & 15;
        if (4 * cx < lw && 4 * cy < lh)
            P.mode_map[(size_t)((W.lcu_y >> 2) + cy) * P.map_pitch + (W.lcu_x >> 2) + cx] = L.mode[0][(cy + 1) * 36 + cx + 1];
    }
    if (P.prof && t == 0) {
        unsigned long long *q = P.prof + 16 * (size_t)((W.lcu_y >> 6) * ((P.width + 63) >> 6) + (W.lcu_x >> 6));
        q[0] = c_pred, q[1] = c_enc, q[2] = __builtin_readcyclecounter() - c0, q[3] = W.num_cus;
        q[8] = c_ph[0], q[9] = c_ph[1], q[10] = c_ph[2], q[11] = c_ph[3]; /* prediction: availability, substitution, smoothing, samples */
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_encode_lcu(EpPicture P, const typename EpTypes<T>::Work *__restrict__ works,
                                                    typename EpTypes<T>::Result *__restrict__ results)
{
    __shared__ EpShared<T> S;
    __shared__ EpLocal<T> L;
    ep_encode_lcu<T>(P, works[blockIdx.x], results[blockIdx.x], S, L);
}

/* AssignEncDecSegments on the device (Codec/EbEncDecProcess.c:1540: an LCU may start when its left and its top-right LCUs are done):
 * ONE launch encodes a whole picture.  A small persistent grid of workgroups draws LCUs as tickets in raster order - every LCU an
 * LCU waits for has a lower ticket, so it is finished or held by a workgroup that is running, and the wait cannot deadlock.  An LCU
 * publishes itself with a device-scope release (its samples, mode map cells and results are then visible to every XCD's L2); a
 * (tickets follow the wavefront's anti-diagonals, so a grid as wide as the wavefront stays busy).  A
 * waiting workgroup polls the flags with RELAXED loads (an acquire per poll would drop the XCD's L2 contents every few hundred
 * cycles and starve the workgroups that do the work - measured: 20x slower) and acquires ONCE before it reads its neighbours.
 * done[] holds the epoch of the call that finished the LCU. */
template <typename T>
__global__ __launch_bounds__(256) void k_encode_picture(EpPicture P, const typename EpTypes<T>::Work *__restrict__ works,
                                                        typename EpTypes<T>::Result *__restrict__ results, int nlcu,
                                                        int wl, unsigned *ticket, unsigned *done, const unsigned *__restrict__ order, unsigned epoch)
{
    __shared__ EpShared<T> S;
    __shared__ EpLocal<T> L;
    __shared__ unsigned s_ticket;
    for (;;) {
        __syncthreads(); /* the previous LCU's readers of s_ticket are through */
        if (threadIdx.x == 0)
            s_ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        if ((int)s_ticket >= nlcu)
            return;
        const int lcu = (int)order[s_ticket];
        const typename EpTypes<T>::Work &W = works[lcu];
        const unsigned long long w0 = P.prof ? __builtin_readcyclecounter() : 0;
        if (threadIdx.x == 0) {
            const int x = W.lcu_x >> 6;
            /* left: (x-1, y); top-right: (x+1, y-1), or the top LCU in the last column of a tile / picture (EbEncDecProcess.c:1585-1640) */
            const int dep0 = W.tile_left ? -1 : lcu - 1;
            const int dep1 = W.tile_top ? -1 : (W.tile_right || x + 1 >= wl) ? lcu - wl : lcu - wl + 1;
            if (dep0 >= 0)
                while (__hip_atomic_load(&done[dep0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch)
                    __builtin_amdgcn_s_sleep(16);
            if (dep1 >= 0)
                while (__hip_atomic_load(&done[dep1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch)
                    __builtin_amdgcn_s_sleep(16);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); /* this CU's / XCD's caches drop what they held of the neighbours' samples */
        }
        __syncthreads();
        const unsigned long long w1 = P.prof ? __builtin_readcyclecounter() : 0;
        ep_encode_lcu<T>(P, W, results[lcu], S, L);
        __syncthreads(); /* every thread's stores of this LCU are issued */
        if (P.prof && threadIdx.x == 0)
            P.prof[16 * (size_t)lcu + 4] = w1 - w0, P.prof[16 * (size_t)lcu + 5] = w0, P.prof[16 * (size_t)lcu + 6] = __builtin_readcyclecounter();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(&done[lcu], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

/* ---- host side ------------------------------------------------------------------------------------------------------------- */
extern "C" int svt_amd_encdec_picture_create(SvtAmdContext *ctx, uint16_t width, uint16_t height, int bytes_per_sample, SvtAmdEncDecPicture **out)
{
    if (!ctx || !out || width < 8 || height < 8 || (width & 7) || (height & 7) || (bytes_per_sample != 1 && bytes_per_sample != 2)) {
        svt_amd_set_error("svt_amd_encdec_picture_create: bad parameter (1 or 2 bytes per sample, dimensions multiples of 8)");
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
    p->nlcu = ((width + 63) / 64) * ((height + 63) / 64);
    if (hipMalloc((void **)&p->d_cost, sizeof(SvtAmdCabacCost)) != hipSuccess || rate_tables_once(ctx->device))
        return SVT_AMD_ERR_RESOURCES;
    p->d.cost = p->d_cost;
    if (hipMalloc((void **)&p->d_sync, sizeof(unsigned) * (size_t)(1 + 2 * p->nlcu)) != hipSuccess)
        return SVT_AMD_ERR_RESOURCES;
    HIP_TRY(hipMemset(p->d_sync, 0, sizeof(unsigned) * (size_t)(1 + p->nlcu)));
    {   /* ticket -> LCU in wavefront order: anti-diagonals x + 2y (the left and the top-right neighbour lie on the diagonal before),
         * so that a grid no wider than the wavefront keeps every workgroup busy */
        const int wl = (width + 63) / 64, hl = (height + 63) / 64;
        std::vector<unsigned> order;
        order.reserve((size_t)p->nlcu);
        for (int d = 0; d <= (wl - 1) + 2 * (hl - 1); d++)
            for (int y = 0; y < hl; y++) {
                const int x = d - 2 * y;
                if (x >= 0 && x < wl)
                    order.push_back((unsigned)(y * wl + x));
            }
        HIP_TRY(hipMemcpy(p->d_sync + 1 + p->nlcu, order.data(), sizeof(unsigned) * order.size(), hipMemcpyHostToDevice));
    }
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
    pic->deblocked = pic->sao_done = false;
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
    if (pic->d_sync)
        (void)hipFree(pic->d_sync);
    if (pic->d.prof)
        (void)hipFree(pic->d.prof);
    if (pic->d_cost)
        (void)hipFree(pic->d_cost);
    for (int k = 0; k < 3; k++) {
        if (pic->dbk[k])
            (void)hipFree(pic->dbk[k]);
        if (pic->fin[k])
            (void)hipFree(pic->fin[k]);
        if (pic->refp[k])
            (void)hipFree(pic->refp[k]);
    }
    free(pic);
    return SVT_AMD_OK;
}

/* P / B pictures: what the inter units of the picture read besides the LCU records - the reference pictures of list 0 / 1 (device
 * memory, whole padded planes; either may be NULL) and pictureControlSetPtr->cabacCost (HOST pointer, copied on the context's stream).
 * Holds until the next call for this picture object. */
extern "C" int svt_amd_encdec_picture_set_inter(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdRefPicture *ref0,
                                                const SvtAmdRefPicture *ref1, const SvtAmdCabacCost *cost)
{
    if (!ctx || !pic || !cost)
        return SVT_AMD_ERR_BAD_PARAM; /* no reference picture at all: an I picture that needs the rate tables (PM-core quantiser) */
    const SvtAmdRefPicture *refs[2] = {ref0, ref1};
    for (int l = 0; l < 2; l++) {
        const SvtAmdRefPicture *r = refs[l];
        pic->has_ref[l] = r != nullptr;
        memset(&pic->d.ref[l], 0, sizeof(pic->d.ref[l]));
        if (!r)
            continue;
        if (!r->d_y || !r->d_cb || !r->d_cr || r->width != pic->d.width || r->height != pic->d.height || r->originX < 8 || r->originY < 8 ||
            r->strideY < r->width + 2 * r->originX || r->strideC < (r->width + 2 * r->originX) / 2) {
            svt_amd_set_error("svt_amd_encdec_picture_set_inter: reference picture %d does not fit the picture", l);
            return SVT_AMD_ERR_BAD_PARAM;
        }
        EpRefPlanes &E = pic->d.ref[l];
        E.plane[0] = r->d_y, E.plane[1] = r->d_cb, E.plane[2] = r->d_cr;
        E.stride[0] = r->strideY, E.stride[1] = r->strideC;
        E.originX = (int32_t)r->originX, E.originY = (int32_t)r->originY, E.width = (int32_t)r->width, E.height = (int32_t)r->height;
        E.size[0] = (int32_t)(r->strideY * (r->height + 2 * r->originY)), E.size[1] = (int32_t)(r->strideC * ((r->height + 2 * r->originY) / 2));
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(pic->d_cost, cost, sizeof(*cost), hipMemcpyHostToDevice, ctx->stream)); /* pageable source: staged before the call returns */
    pic->has_cost = true;
    return SVT_AMD_OK;
}

/* hip_encdec_segment: works / results are HOST arrays of n LCUs that do not depend on each other (one wavefront step); blocking.
 * Contexts (lanes) may call concurrently for different LCUs of the same picture as long as the wavefront order holds between calls. */
template <typename WorkT>
static int ep_validate(const SvtAmdEncDecPicture *pic, const WorkT *works, int n, const char *who)
{
    for (int i = 0; i < n; i++) {
        if (works[i].num_cus > SVT_AMD_LCU_MAX_CUS || works[i].lcu_x >= pic->d.width || works[i].lcu_y >= pic->d.height || (works[i].lcu_x & 63) ||
            (works[i].lcu_y & 63)) {
            svt_amd_set_error("%s: bad LCU %d", who, i);
            return SVT_AMD_ERR_BAD_PARAM;
        }
        if (works[i].pm_core && !pic->has_cost) {
            svt_amd_set_error("%s: LCU %d asks for the PM-core quantiser without the picture's rate tables (svt_amd_encdec_picture_set_inter)", who, i);
            return SVT_AMD_ERR_BAD_PARAM;
        }
        for (int c = 0; c < works[i].num_cus; c++) {
            const SvtAmdLcuCu &u = works[i].cu[c];
            const bool inter = u.pred_mode == 1;
            if ((u.pred_mode != 2 && !inter) || !(u.size == 8 || u.size == 16 || u.size == 32 || (inter && u.size == 64)) || u.intra_luma_mode > 34 ||
                (u.x & (u.size - 1)) || (u.y & (u.size - 1)) || u.x + u.size > 64 || u.y + u.size > 64 || works[i].lcu_x + u.x + u.size > pic->d.width ||
                works[i].lcu_y + u.y + u.size > pic->d.height) {
                svt_amd_set_error("%s: LCU %d unit %d is not an intra 2Nx2N unit of 8..32 or an inter 2Nx2N unit of 8..64 inside the picture", who, i, c);
                return SVT_AMD_ERR_BAD_PARAM;
            }
            if (inter && (u.inter_dir > 2 || u.inter_kind > SVT_AMD_EP_INTER_SKIP || (u.inter_dir != 1 && !pic->has_ref[0]) || (u.inter_dir != 0 && !pic->has_ref[1]))) {
                svt_amd_set_error("%s: LCU %d unit %d: inter unit without its reference picture (svt_amd_encdec_picture_set_inter) or with a bad direction / kind",
                                  who, i, c);
                return SVT_AMD_ERR_BAD_PARAM;
            }
        }
    }
    return SVT_AMD_OK;
}

template <typename T>
static int encode_lcus(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const typename EpTypes<T>::Work *works, int n, typename EpTypes<T>::Result *results,
                       const char *who)
{
    typedef typename EpTypes<T>::Work WorkT;
    typedef typename EpTypes<T>::Result ResultT;
    if (!ctx || !pic || !works || !results || n < 1 || n > 1024)
        return SVT_AMD_ERR_BAD_PARAM;
    if (pic->d.bps != sizeof(T)) {
        svt_amd_set_error("%s: the picture holds %u-byte samples", who, pic->d.bps);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    int rv = ep_validate(pic, works, n, who);
    if (rv)
        return rv;
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *d = nullptr;
    const size_t wb = sizeof(WorkT) * (size_t)n, rb = sizeof(ResultT) * (size_t)n, wba = (wb + 255) & ~(size_t)255;
    int rc = svt_amd_ctx_scratch(ctx, wba + rb, &d);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(d, works, wb, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_encode_lcu<T>, dim3((unsigned)n), dim3(256), 0, ctx->stream, pic->d, (const WorkT *)d, (ResultT *)(d + wba));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(results, d + wba, rb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_encode_lcus(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works, int n, SvtAmdLcuResult *results)
{
    return encode_lcus<uint8_t>(ctx, pic, works, n, results, "svt_amd_encode_lcus");
}
extern "C" int svt_amd_encode_lcus16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works, int n, SvtAmdLcuResult16 *results)
{
    return encode_lcus<uint16_t>(ctx, pic, works, n, results, "svt_amd_encode_lcus16");
}

/* One call per picture: works / results are HOST arrays of ALL LCUs of the picture in raster order; the wavefront runs on the
 * device (k_encode_picture).  d_works / d_results (optional, device) replace the host arrays: nothing crosses PCIe then. */
template <typename T>
static int encode_picture(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const typename EpTypes<T>::Work *works, typename EpTypes<T>::Result *results,
                          const typename EpTypes<T>::Work *d_works, typename EpTypes<T>::Result *d_results, int parallel_tiles)
{
    typedef typename EpTypes<T>::Work WorkT;
    typedef typename EpTypes<T>::Result ResultT;
    if (pic->d.bps != sizeof(T)) {
        svt_amd_set_error("svt_amd_encode_picture: the picture holds %u-byte samples", pic->d.bps);
        return SVT_AMD_ERR_BAD_PARAM;
    }
    const int n = pic->nlcu, wl = (pic->d.width + 63) / 64;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!d_works) {
        for (int i = 0; i < n; i++)
            if (works[i].lcu_x != (i % wl) * 64 || works[i].lcu_y != (i / wl) * 64) {
                svt_amd_set_error("svt_amd_encode_picture: LCU %d is not at raster position %d", i, i);
                return SVT_AMD_ERR_BAD_PARAM;
            }
        int rv = ep_validate(pic, works, n, "svt_amd_encode_picture");
        if (rv)
            return rv;
        parallel_tiles = 0; /* tiles = LCUs that wait for nobody (top-left corners) */
        for (int i = 0; i < n; i++)
            parallel_tiles += works[i].tile_left && works[i].tile_top;
        uint8_t *d = nullptr;
        const size_t wb = sizeof(WorkT) * (size_t)n, rb = sizeof(ResultT) * (size_t)n, wba = (wb + 255) & ~(size_t)255;
        int rc = svt_amd_ctx_scratch(ctx, wba + rb, &d);
        if (rc)
            return rc;
        HIP_TRY(hipMemcpyAsync(d, works, wb, hipMemcpyHostToDevice, ctx->stream));
        d_works = (const WorkT *)d, d_results = (ResultT *)(d + wba);
    }
    pic->epoch++;
    pic->deblocked = pic->sao_done = false;
    HIP_TRY(hipMemsetAsync(pic->d_sync, 0, sizeof(unsigned), ctx->stream));              /* ticket counter */
    HIP_TRY(hipMemsetAsync(pic->d.mode_map, 0xFF, pic->map_bytes, ctx->stream));          /* nothing coded yet */
    /* persistent grid = the widest wavefront (an LCU row advances two LCUs behind the row above) of every tile that can run on its
     * own: more workgroups would only poll, and they would hold the CUs other pictures' launches could use */
    const int hl = (pic->d.height + 63) / 64;
    int grid = ((wl + 1) / 2 < hl ? (wl + 1) / 2 : hl) * (parallel_tiles > 0 ? parallel_tiles : 1) + 1;
    grid = grid > n ? n : grid > 512 ? 512 : grid;
    hipLaunchKernelGGL(k_encode_picture<T>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, pic->d, d_works, d_results, n, wl, pic->d_sync, pic->d_sync + 1, pic->d_sync + 1 + n, pic->epoch);
    HIP_TRY(hipGetLastError());
    if (results)
        HIP_TRY(hipMemcpyAsync(results, d_results, sizeof(ResultT) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    return SVT_AMD_OK;
}

extern "C" int svt_amd_encode_picture(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works, SvtAmdLcuResult *results)
{
    if (!ctx || !pic || !works || !results)
        return SVT_AMD_ERR_BAD_PARAM;
    int rc = encode_picture<uint8_t>(ctx, pic, works, results, nullptr, nullptr, 0);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_encode_picture16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works, SvtAmdLcuResult16 *results)
{
    if (!ctx || !pic || !works || !results)
        return SVT_AMD_ERR_BAD_PARAM;
    int rc = encode_picture<uint16_t>(ctx, pic, works, results, nullptr, nullptr, 0);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}

/* Device-resident form: work and result arrays already in HBM (what a device-side mode decision would leave there); asynchronous on
 * the context's stream.  The caller vouches for the unit lists (no host copy to validate). */
extern "C" int svt_amd_encode_picture_device(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *d_works, SvtAmdLcuResult *d_results,
                                             int tiles)
{
    if (!ctx || !pic || !d_works || !d_results || tiles < 1)
        return SVT_AMD_ERR_BAD_PARAM;
    return encode_picture<uint8_t>(ctx, pic, nullptr, nullptr, d_works, d_results, tiles);
}

/* ---- deblocking behind the encode pass ------------------------------------------------------------------------------------------ */
/* Once every LCU of the picture is encoded (svt_amd_encode_picture, or LCU by LCU), the device picture goes through the
 * picture-level boundary-strength and deblocking kernels (filter_kernels.hip, proven on recorded pictures) IN PLACE: what the
 * reference's reconstruction holds after its per-LCU drivers - the finished picture when SAO is off.  The two maps those kernels
 * read (coding unit per 8x8 block, luma cbf per 4x4 block) and the QP array come from the contract records the host already has. */
template <typename T>
static int picture_deblock(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const typename EpTypes<T>::Work *works, const typename EpTypes<T>::Result *results,
                           const SvtAmdDeblockParams *prm, void *out_y, void *out_cb, void *out_cr)
{
    if (!ctx || !pic || !works || !results || !prm || pic->d.bps != sizeof(T))
        return SVT_AMD_ERR_BAD_PARAM;
    const uint32_t w = pic->d.width, h = pic->d.height, wl = (w + 63) / 64, nlcu = (uint32_t)pic->nlcu;
    const uint32_t w8 = w / 8, h8 = h / 8, w4 = w / 4, h4 = h / 4;
    std::vector<SvtAmdCuMapEntry> map((size_t)w8 * h8);
    std::vector<uint8_t> cbf((size_t)w4 * h4), qp((size_t)w8 * h8), edge(nlcu);
    ::memset(map.data(), 0, map.size() * sizeof(SvtAmdCuMapEntry));
    for (uint32_t i = 0; i < nlcu; i++) {
        const auto &W = works[i];
        if (W.lcu_x != (i % wl) * 64 || W.lcu_y != (i / wl) * 64) {
            svt_amd_set_error("svt_amd_encdec_picture_deblock: LCU %u is not at raster position %u", i, i);
            return SVT_AMD_ERR_BAD_PARAM;
        }
        edge[i] = (uint8_t)((W.tile_left ? 1 : 0) | (W.tile_top ? 2 : 0));
        for (int c = 0; c < W.num_cus; c++) {
            const SvtAmdLcuCu &u = W.cu[c];
            const uint32_t x0 = W.lcu_x + u.x, y0 = W.lcu_y + u.y;
            if (x0 + u.size > w || y0 + u.size > h || u.size < 8)
                return SVT_AMD_ERR_BAD_PARAM;
            SvtAmdCuMapEntry e;
            ::memset(&e, 0, sizeof(e));
            e.mode = u.pred_mode, e.size_log2 = (uint8_t)(u.size == 8 ? 3 : u.size == 16 ? 4 : u.size == 32 ? 5 : 6);
            if (u.pred_mode == 1) /* inter: the prediction unit's direction and motion vectors decide the strength of its edges */
                e.dir = u.inter_dir, ::memcpy(e.mv, u.mv, sizeof(e.mv));
            for (uint32_t y = y0 / 8; y < (y0 + u.size) / 8; y++)
                for (uint32_t x = x0 / 8; x < (x0 + u.size) / 8; x++)
                    map[(size_t)y * w8 + x] = e, qp[(size_t)y * w8 + x] = u.qp;
            if (u.size == 64) { /* four 32x32 transform units: result entries c + 1 .. c + 4 */
                for (uint32_t y = 0; y < 16; y++)
                    for (uint32_t t = 0; t < 2; t++)
                        ::memset(&cbf[(size_t)(y0 / 4 + y) * w4 + x0 / 4 + 8 * t], results[i].cu[c + 1 + 2 * (y >> 3) + t].cbf[0], 8);
            } else {
                for (uint32_t y = y0 / 4; y < (y0 + u.size) / 4; y++)
                    ::memset(&cbf[(size_t)y * w4 + x0 / 4], results[i].cu[c].cbf[0], u.size / 4);
            }
        }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t b_map = (map.size() * sizeof(SvtAmdCuMapEntry) + 255) & ~(size_t)255, b_cbf = (cbf.size() + 255) & ~(size_t)255,
                 b_qp = (qp.size() + 255) & ~(size_t)255, b_edge = ((size_t)nlcu + 255) & ~(size_t)255, b_bs = (size_t)nlcu * 256;
    uint8_t *d = nullptr;
    int rc = svt_amd_ctx_scratch(ctx, b_map + b_cbf + b_qp + b_edge + 2 * b_bs, &d);
    if (rc)
        return rc;
    uint8_t *d_map = d, *d_cbf = d_map + b_map, *d_qp = d_cbf + b_cbf, *d_edge = d_qp + b_qp, *d_bsv = d_edge + b_edge, *d_bsh = d_bsv + b_bs;
    HIP_TRY(hipMemcpyAsync(d_map, map.data(), map.size() * sizeof(SvtAmdCuMapEntry), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_cbf, cbf.data(), cbf.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_qp, qp.data(), qp.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_edge, edge.data(), edge.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream)); /* the host vectors back the copies */
    if ((rc = svt_amd_bs_picture(ctx, (const SvtAmdCuMapEntry *)d_map, d_cbf, w, h, prm->slice_type, prm->ref_poc[0], prm->ref_poc[1], d_edge, d_bsv,
                                 d_bsh)) != 0)
        return rc;
    if (pic->d.pitch[1] != pic->d.pitch[2])
        return SVT_AMD_ERR_BAD_PARAM;
    /* the deblocked picture is a second set of planes: the un-deblocked one stays (the neighbours of LCUs still to come, and the part of
     * the SAO statistics that the reference gathers before an LCU's right / bottom edges are filtered) */
    for (int k = 0; k < 3; k++) {
        if (!pic->dbk[k] && hipMalloc((void **)&pic->dbk[k], pic->plane_bytes[k]) != hipSuccess) {
            svt_amd_set_error("hipMalloc (deblocked picture) failed");
            return SVT_AMD_ERR_RESOURCES;
        }
        HIP_TRY(hipMemcpyAsync(pic->dbk[k], pic->d.rec[k], pic->plane_bytes[k], hipMemcpyDeviceToDevice, ctx->stream));
    }
    if ((rc = svt_amd_dlf_picture(ctx, (int)sizeof(T), pic->dbk[0], pic->d.pitch[0], pic->dbk[1], pic->dbk[2], pic->d.pitch[1], w, h, d_bsv, d_bsh,
                                  d_qp, w8, prm->tc_offset, prm->beta_offset, prm->cb_qp_offset, prm->cr_qp_offset)) != 0)
        return rc;
    pic->deblocked = true;
    void *outs[3] = {out_y, out_cb, out_cr};
    for (int k = 0; k < 3; k++)
        if (outs[k]) {
            const uint32_t pw = k ? w / 2 : w, ph = k ? h / 2 : h;
            HIP_TRY(hipMemcpy2DAsync(outs[k], (size_t)pw * sizeof(T), pic->dbk[k], (size_t)pic->d.pitch[k] * sizeof(T), (size_t)pw * sizeof(T), ph,
                                     hipMemcpyDeviceToHost, ctx->stream));
        }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}

/* ---- SAO behind the deblocked device picture ------------------------------------------------------------------------------------ */
/* the LCUs' source samples as planes (the statistics kernels read planes) */
template <typename T>
__global__ __launch_bounds__(256) void k_ep_source_planes(const typename EpTypes<T>::Work *__restrict__ works, T *sy, T *scb, T *scr, int pitchY, int pitchC,
                                                          int width, int height)
{
    const typename EpTypes<T>::Work &W = works[blockIdx.x];
    const int lw = min(64, width - (int)W.lcu_x), lh = min(64, height - (int)W.lcu_y);
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int y = i >> 6, x = i & 63;
        if (x < lw && y < lh)
            sy[(size_t)(W.lcu_y + y) * pitchY + W.lcu_x + x] = W.src_y[i];
    }
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int y = i >> 5, x = i & 31;
        if (x < lw / 2 && y < lh / 2) {
            scb[(size_t)(W.lcu_y / 2 + y) * pitchC + W.lcu_x / 2 + x] = W.src_cb[i];
            scr[(size_t)(W.lcu_y / 2 + y) * pitchC + W.lcu_x / 2 + x] = W.src_cr[i];
        }
    }
}

/* The picture as the reference's SaoGenerationDecision sees each LCU (EbCodingLoop.c:4600-4750): the LCU's own deblocking drivers have run,
 * those of the LCUs to its right and below have not.  Those later drivers own the 8x8 filter blocks centred on the LCU boundary
 * (LCUBoundaryDLFCore, EbDeblockingFilter.c:2828), i.e. every edge segment inside the last 4 columns / rows of the LCU (in the plane's own
 * samples) and nothing else inside it: the LCU is the deblocked picture with those strips still un-deblocked, where a neighbour OF THE SAME
 * TILE follows (at picture and tile edges the LCU's own LCUPictureEdgeDLFCore has finished them; follow: bit 0 right, bit 1 below).
 * (tests/test_oracle_encodepass_golden.py::test_encoder_order_sao_statistics_from_two_pictures proves it on the encoder's own records.) */
template <typename T>
__global__ __launch_bounds__(256) void k_ep_sao_composite(const T *__restrict__ dbk, const T *__restrict__ rec, T *__restrict__ out, int pitch, int w, int h,
                                                          int lg_lcu, int wl, const uint8_t *__restrict__ follow)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, lcu = 1 << lg_lcu;
    if (x >= w)
        return;
    const int f = follow[(y >> lg_lcu) * wl + (x >> lg_lcu)];
    const bool right = (x & (lcu - 1)) >= lcu - 4 && (f & 1), bottom = (y & (lcu - 1)) >= lcu - 4 && (f & 2);
    const size_t o = (size_t)y * pitch + x;
    out[o] = (right || bottom) ? rec[o] : dbk[o];
}

/* Statistics of every LCU in the encoder's order, the parameter decision of the whole picture (merge wavefront) and the application, behind
 * svt_amd_encdec_picture_deblock: what is left in the picture object (and copied out) is the encoder's finished reconstruction. */
template <typename T>
static int picture_sao(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const typename EpTypes<T>::Work *works, const SvtAmdSaoDecisionParams *prm,
                       const uint8_t *enable, SvtAmdSaoLcuParams *lcu_out, void *out_y, void *out_cb, void *out_cr, bool apply)
{
    typedef typename EpTypes<T>::Work WorkT;
    if (!ctx || !pic || !works || !prm || pic->d.bps != sizeof(T))
        return SVT_AMD_ERR_BAD_PARAM;
    if (!pic->deblocked && apply) {
        svt_amd_set_error("svt_amd_encdec_picture_sao: svt_amd_encdec_picture_deblock comes first");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    const uint32_t w = pic->d.width, h = pic->d.height, wl = (w + 63) / 64, hl = (h + 63) / 64, nlcu = (uint32_t)pic->nlcu;
    std::vector<SvtAmdSaoLcuParams> lp(nlcu);
    std::vector<uint8_t> follow(nlcu);
    ::memset(lp.data(), 0, nlcu * sizeof(SvtAmdSaoLcuParams));
    for (uint32_t i = 0; i < nlcu; i++) {
        if (works[i].lcu_x != (i % wl) * 64 || works[i].lcu_y != (i / wl) * 64)
            return SVT_AMD_ERR_BAD_PARAM;
        const bool bottom_edge = i + wl >= nlcu || works[i + wl].tile_top;
        lp[i].edge_flags = (uint8_t)((works[i].tile_left ? 1 : 0) | (works[i].tile_right ? 2 : 0) | (works[i].tile_top ? 4 : 0) | (bottom_edge ? 8 : 0));
        follow[i] = (uint8_t)(((i % wl) + 1 < wl && !works[i].tile_right ? 1 : 0) | (!bottom_edge ? 2 : 0));
    }
    HIP_TRY(hipSetDevice(ctx->device));
    for (int k = 0; k < 3 && apply; k++)
        if (!pic->fin[k] && hipMalloc((void **)&pic->fin[k], pic->plane_bytes[k]) != hipSuccess)
            return SVT_AMD_ERR_RESOURCES;
    auto up = [](size_t n) { return (n + 255) & ~(size_t)255; };
    const size_t b_works = up(sizeof(WorkT) * nlcu), b_pl[3] = {up(pic->plane_bytes[0]), up(pic->plane_bytes[1]), up(pic->plane_bytes[2])};
    const size_t b_stats = up(sizeof(SvtAmdSaoStats) * nlcu), b_par = up(sizeof(SvtAmdSaoLcuParams) * nlcu), b_cost = up(16 * (size_t)nlcu), b_en = up(nlcu);
    uint8_t *d = nullptr;
    int rc = svt_amd_ctx_scratch(ctx, b_works + 2 * (b_pl[0] + b_pl[1] + b_pl[2]) + 3 * b_stats + b_par + b_cost + 2 * b_en, &d);
    if (rc)
        return rc;
    uint8_t *d_works = d, *d_src[3], *d_cmp[3], *q = d + b_works;
    for (int k = 0; k < 3; k++)
        d_src[k] = q, q += b_pl[k];
    for (int k = 0; k < 3; k++)
        d_cmp[k] = q, q += b_pl[k];
    SvtAmdSaoStats *d_stats[3];
    for (int k = 0; k < 3; k++)
        d_stats[k] = (SvtAmdSaoStats *)q, q += b_stats;
    SvtAmdSaoLcuParams *d_par = (SvtAmdSaoLcuParams *)q;
    q += b_par;
    int64_t *d_cost = (int64_t *)q;
    q += b_cost;
    uint8_t *d_en = q, *d_follow = q + b_en;
    HIP_TRY(hipMemcpyAsync(d_works, works, sizeof(WorkT) * nlcu, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_follow, follow.data(), nlcu, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_par, lp.data(), sizeof(SvtAmdSaoLcuParams) * nlcu, hipMemcpyHostToDevice, ctx->stream));
    if (enable)
        HIP_TRY(hipMemcpyAsync(d_en, enable, nlcu, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_stats[0], 0, 3 * b_stats, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream)); /* lp backs its copy */
    const int pY = (int)pic->d.pitch[0], pC = (int)pic->d.pitch[1];
    hipLaunchKernelGGL(k_ep_source_planes<T>, dim3(nlcu), dim3(256), 0, ctx->stream, (const WorkT *)d_works, (T *)d_src[0], (T *)d_src[1], (T *)d_src[2], pY, pC,
                       (int)w, (int)h);
    for (int k = 0; k < 3 && pic->deblocked; k++) {
        const int pw = k ? w / 2 : w, ph = k ? h / 2 : h;
        hipLaunchKernelGGL(k_ep_sao_composite<T>, dim3((pw + 255) / 256, ph), dim3(256), 0, ctx->stream, (const T *)pic->dbk[k], (const T *)pic->d.rec[k],
                           (T *)d_cmp[k], k ? pC : pY, pw, ph, k ? 5 : 6, (int)wl, (const uint8_t *)d_follow);
    }
    HIP_TRY(hipGetLastError());
    /* GatherSaoStatisticsLcu* of the components the mode looks at (EbSampleAdaptiveOffsetGenerationDecision.c:647-760) */
    const int ncomp = prm->mm_sao ? 3 : (prm->temporal_layer < 2 ? 1 : 0);
    for (int k = 0; k < ncomp; k++)
        if ((rc = svt_amd_sao_gather_picture(ctx, (int)sizeof(T), d_src[k], k ? pC : pY, pic->deblocked ? (const void *)d_cmp[k] : (const void *)pic->d.rec[k],
                                             k ? pC : pY, k ? w / 2 : w, k ? h / 2 : h, k ? 32 : 64, prm->mm_sao ? 0 : 1, d_stats[k])) != 0)
            return rc; /* a picture that is never deblocked (decide-only call) is its own encoder-order view */
    if ((rc = svt_amd_sao_decide_picture(ctx, prm, d_stats[0], d_stats[1], d_stats[2], wl, hl, enable ? d_en : nullptr, d_par, d_cost)) != 0)
        return rc;
    const void *srcs[3] = {pic->dbk[0], pic->dbk[1], pic->dbk[2]};
    void *dsts[3] = {pic->fin[0], pic->fin[1], pic->fin[2]};
    if (apply && (rc = svt_amd_sao_apply_picture(ctx, (int)sizeof(T), srcs, dsts, pic->d.pitch[0], pic->d.pitch[1], w, h, d_par, 1, 1)) != 0)
        return rc;
    if (lcu_out)
        HIP_TRY(hipMemcpyAsync(lcu_out, d_par, sizeof(SvtAmdSaoLcuParams) * nlcu, hipMemcpyDeviceToHost, ctx->stream));
    void *outs[3] = {out_y, out_cb, out_cr};
    for (int k = 0; k < 3 && apply; k++)
        if (outs[k]) {
            const uint32_t pw = k ? w / 2 : w, ph = k ? h / 2 : h;
            HIP_TRY(hipMemcpy2DAsync(outs[k], (size_t)pw * sizeof(T), pic->fin[k], (size_t)pic->d.pitch[k] * sizeof(T), (size_t)pw * sizeof(T), ph,
                                     hipMemcpyDeviceToHost, ctx->stream));
        }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    pic->sao_done = pic->sao_done || apply;
    return SVT_AMD_OK;
}

/* PadRefAndSetFlags (Codec/EbEncDecProcess.c:1805; GeneratePadding / GeneratePadding16Bit): the finished picture inside a frame of
 * replicated edge samples */
template <typename T>
__global__ __launch_bounds__(256) void k_ep_pad(const T *__restrict__ src, int spitch, int w, int h, T *__restrict__ dst, int dstride, int ox, int oy)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w + 2 * ox)
        return;
    dst[(size_t)y * dstride + x] = src[(size_t)min(max(y - oy, 0), h - 1) * spitch + min(max(x - ox, 0), w - 1)];
}

/* The picture object's latest stage (after SAO, else deblocked, else as encoded) as a padded reference picture in HBM: what
 * svt_amd_encdec_picture_set_inter of a later picture takes - reference pictures never leave the device.  origin_x / origin_y: luma padding
 * (the reference uses LCU size + 16 = 80); the planes are (width + 2 origin_x) samples wide, chroma half of everything.  out_*: optional
 * HOST copies of the padded planes.  The planes live until the picture object is destroyed or the call is repeated. */
extern "C" int svt_amd_encdec_picture_reference(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, uint32_t origin_x, uint32_t origin_y, SvtAmdRefPicture *ref,
                                                void *out_y, void *out_cb, void *out_cr)
{
    if (!ctx || !pic || !ref || origin_x < 8 || origin_y < 8 || (origin_x & 1) || (origin_y & 1) || origin_x > 256 || origin_y > 256)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t w = pic->d.width, h = pic->d.height, bps = pic->d.bps;
    uint8_t *const *stage = pic->sao_done ? pic->fin : pic->deblocked ? pic->dbk : pic->d.rec;
    void *outs[3] = {out_y, out_cb, out_cr};
    for (int k = 0; k < 3; k++) {
        const int sh = k ? 1 : 0, pw = (int)(w >> sh), ph = (int)(h >> sh), ox = (int)(origin_x >> sh), oy = (int)(origin_y >> sh), stride = pw + 2 * ox, rows = ph + 2 * oy;
        const size_t need = (size_t)stride * rows * bps;
        if (pic->refp_bytes[k] < need) {
            if (pic->refp[k])
                HIP_TRY(hipFree(pic->refp[k]));
            pic->refp[k] = nullptr, pic->refp_bytes[k] = 0;
            if (hipMalloc((void **)&pic->refp[k], need) != hipSuccess)
                return SVT_AMD_ERR_RESOURCES;
            pic->refp_bytes[k] = need;
        }
        if (bps == 1)
            hipLaunchKernelGGL(k_ep_pad<uint8_t>, dim3((stride + 255) / 256, rows), dim3(256), 0, ctx->stream, (const uint8_t *)stage[k], (int)pic->d.pitch[k], pw, ph,
                               (uint8_t *)pic->refp[k], stride, ox, oy);
        else
            hipLaunchKernelGGL(k_ep_pad<uint16_t>, dim3((stride + 255) / 256, rows), dim3(256), 0, ctx->stream, (const uint16_t *)stage[k], (int)pic->d.pitch[k], pw,
                               ph, (uint16_t *)pic->refp[k], stride, ox, oy);
        if (outs[k])
            HIP_TRY(hipMemcpyAsync(outs[k], pic->refp[k], need, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream)); /* other lanes' pictures may read the planes right away */
    ref->d_y = pic->refp[0], ref->d_cb = pic->refp[1], ref->d_cr = pic->refp[2];
    ref->strideY = w + 2 * origin_x, ref->strideC = (w >> 1) + 2 * (origin_x >> 1), ref->originX = origin_x, ref->originY = origin_y;
    ref->width = w, ref->height = h;
    return SVT_AMD_OK;
}
extern "C" int svt_amd_encdec_picture_sao(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works, const SvtAmdSaoDecisionParams *params,
                                          const uint8_t *enable, SvtAmdSaoLcuParams *lcu_params, uint8_t *out_y, uint8_t *out_cb, uint8_t *out_cr)
{
    return picture_sao<uint8_t>(ctx, pic, works, params, enable, lcu_params, out_y, out_cb, out_cr, true);
}
/* the parameter decision alone, on the picture object as it stands (deblocked: the encoder-order view; not deblocked: the picture as encoded -
 * allowEncDecMismatch pictures, whose parameters the reference decides on its un-deblocked reconstruction and signals without applying them) */
extern "C" int svt_amd_encdec_picture_sao_decide(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const void *works, const SvtAmdSaoDecisionParams *params,
                                                 const uint8_t *enable, SvtAmdSaoLcuParams *lcu_params)
{
    if (!pic || !lcu_params)
        return SVT_AMD_ERR_BAD_PARAM;
    return pic->d.bps == 1 ? picture_sao<uint8_t>(ctx, pic, (const SvtAmdLcuWork *)works, params, enable, lcu_params, nullptr, nullptr, nullptr, false)
                           : picture_sao<uint16_t>(ctx, pic, (const SvtAmdLcuWork16 *)works, params, enable, lcu_params, nullptr, nullptr, nullptr, false);
}
extern "C" int svt_amd_encdec_picture_sao16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works, const SvtAmdSaoDecisionParams *params,
                                            const uint8_t *enable, SvtAmdSaoLcuParams *lcu_params, uint16_t *out_y, uint16_t *out_cb, uint16_t *out_cr)
{
    return picture_sao<uint16_t>(ctx, pic, works, params, enable, lcu_params, out_y, out_cb, out_cr, true);
}
extern "C" int svt_amd_encdec_picture_deblock(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork *works, const SvtAmdLcuResult *results,
                                              const SvtAmdDeblockParams *params, uint8_t *out_y, uint8_t *out_cb, uint8_t *out_cr)
{
    return picture_deblock<uint8_t>(ctx, pic, works, results, params, out_y, out_cb, out_cr);
}
extern "C" int svt_amd_encdec_picture_deblock16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuWork16 *works,
                                                const SvtAmdLcuResult16 *results, const SvtAmdDeblockParams *params, uint16_t *out_y, uint16_t *out_cb,
                                                uint16_t *out_cr)
{
    return picture_deblock<uint16_t>(ctx, pic, works, results, params, out_y, out_cb, out_cr);
}

/* ---- LCUs encoded by the host: their last row / column and edge mode types enter the device picture --------------------------- */
template <typename T>
__global__ __launch_bounds__(256) void k_put_borders(EpPicture P, const typename EpTypes<T>::Border *B)
{
    const typename EpTypes<T>::Border &b = B[blockIdx.x];
    const int t = threadIdx.x;
    const int lw = min(64, (int)P.width - (int)b.lcu_x), lh = min(64, (int)P.height - (int)b.lcu_y);
    /* 0..63 bottom Y, 64..127 right Y, 128..159 / 160..191 bottom / right Cb, 192..223 / 224..255 Cr */
    const int p = t < 128 ? 0 : (t < 192 ? 1 : 2), e = p == 0 ? t : (p == 1 ? t - 128 : t - 192), n = p ? 32 : 64;
    const bool right = e >= n;
    const int i = right ? e - n : e, w = p ? lw >> 1 : lw, h = p ? lh >> 1 : lh;
    const int x0 = p ? b.lcu_x >> 1 : b.lcu_x, y0 = p ? b.lcu_y >> 1 : b.lcu_y;
    const T *src = p == 0 ? (right ? b.right_y : b.bottom_y) : p == 1 ? (right ? b.right_cb : b.bottom_cb) : (right ? b.right_cr : b.bottom_cr);
    if (i < (right ? h : w))
        ((T *)P.rec[p])[right ? (size_t)(y0 + i) * P.pitch[p] + x0 + w - 1 : (size_t)(y0 + h - 1) * P.pitch[p] + x0 + i] = src[i];
    if (t < 16 && t < (lw >> 2))
        P.mode_map[(size_t)((b.lcu_y + lh - 1) >> 2) * P.map_pitch + (b.lcu_x >> 2) + t] = b.mode_bottom[t];
    if (t >= 16 && t < 32 && t - 16 < (lh >> 2))
        P.mode_map[(size_t)((b.lcu_y >> 2) + t - 16) * P.map_pitch + ((b.lcu_x + lw - 1) >> 2)] = b.mode_right[t - 16];
}

template <typename T>
static int put_borders(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const typename EpTypes<T>::Border *borders, int n)
{
    typedef typename EpTypes<T>::Border BorderT;
    if (!ctx || !pic || !borders || n < 1 || n > 4096 || pic->d.bps != sizeof(T))
        return SVT_AMD_ERR_BAD_PARAM;
    for (int i = 0; i < n; i++)
        if (borders[i].lcu_x >= pic->d.width || borders[i].lcu_y >= pic->d.height || ((borders[i].lcu_x | borders[i].lcu_y) & 63)) {
            svt_amd_set_error("svt_amd_encdec_picture_put_borders: bad LCU %d", i);
            return SVT_AMD_ERR_BAD_PARAM;
        }
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *d = nullptr;
    int rc = svt_amd_ctx_scratch(ctx, sizeof(BorderT) * (size_t)n, &d);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(d, borders, sizeof(BorderT) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_put_borders<T>, dim3((unsigned)n), dim3(256), 0, ctx->stream, pic->d, (const BorderT *)d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}
extern "C" int svt_amd_encdec_picture_put_borders(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuBorder *borders, int n)
{
    return put_borders<uint8_t>(ctx, pic, borders, n);
}
extern "C" int svt_amd_encdec_picture_put_borders16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdLcuBorder16 *borders, int n)
{
    return put_borders<uint16_t>(ctx, pic, borders, n);
}

/* debug: out == NULL arms the per-LCU clock sums (16 x u64 per LCU: prediction, encode, copy-out, units, wait, start, end, -, then the
 * prediction's four sub-phases),
 * a later call with a HOST buffer fetches them */
extern "C" int svt_amd_debug_encdec_profile(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, unsigned long long *out)
{
    if (!ctx || !pic)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = sizeof(unsigned long long) * 16 * (size_t)pic->nlcu;
    if (!pic->d.prof) {
        HIP_TRY(hipMalloc((void **)&pic->d.prof, bytes));
        HIP_TRY(hipMemset(pic->d.prof, 0, bytes));
    }
    if (out) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpy(out, pic->d.prof, bytes, hipMemcpyDeviceToHost));
    }
    return SVT_AMD_OK;
}
