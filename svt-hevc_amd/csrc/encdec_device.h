/* encdec_device.h - the device side of the device-resident encode pass (encdec_kernels.hip) as a header, so that the mode-decision kernel
 * (md_kernels.hip) can run an LCU's encode pass right behind its mode decision in the same workgroup: picture state, the LCU in LDS, the
 * intra / inter prediction of a plane by one wave, the fused encode unit and the coding-unit loop of one LCU (ep_encode_lcu). */
#ifndef SVT_AMD_ENCDEC_DEVICE_H
#define SVT_AMD_ENCDEC_DEVICE_H
#include "txfm_device.h"
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
    unsigned *d_order_rect; /* ticket order of a rank's rectangle (svt_amd_encode_picture_rect), allocated on first use */
    unsigned *d_order_md;   /* the same for the rectangle set with svt_amd_encdec_picture_set_rect (mode-decision calls) */
    int md_rect_n;          /* LCUs of that rectangle, 0 = the whole picture */
    SvtAmdRect md_rect;
    unsigned epoch;
    int nlcu;
    SvtAmdCabacCost *d_cost;
    SvtAmdCabacCost *h_cost;   /* page-locked staging of the caller's (pageable) rate tables: the copy to d_cost stays asynchronous */
    bool has_ref[2], has_cost;
    /* the in-loop filters behind the encode pass: the deblocked picture and the picture after SAO live beside the un-deblocked one (the SAO
     * statistics need both, svt_amd_encdec_picture_sao); same pitches as rec[] */
    uint8_t *dbk[3], *fin[3];
    bool deblocked, sao_done;
    /* the finished picture with its padding: a reference picture of later pictures (svt_amd_encdec_picture_reference) */
    uint8_t *refp[3];
    size_t refp_bytes[3];
    /* the mode decision's part (md_kernels.hip): neighbour maps, source planes, per-LCU arrays; allocated by the first svt_amd_md_encode_picture */
    struct SvtAmdMdState *md;
    /* recorded behind every picture-level call that writes the picture's stages (encode pass, mode decision + encode pass, deblocking, SAO, an exchange that fills it):
     * a reader on another context's stream - svt_amd_encdec_picture_import - orders itself behind it; `written`: the object holds an encoded picture at all */
    hipEvent_t ev_written;
    bool written;
};
/* the picture's stages are (being) written by work queued on ctx's stream */
static inline int ep_picture_written(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic)
{
    if (!pic->ev_written && hipEventCreateWithFlags(&pic->ev_written, hipEventDisableTiming) != hipSuccess) {
        pic->ev_written = nullptr;
        return SVT_AMD_ERR_RESOURCES;
    }
    if (hipEventRecord(pic->ev_written, ctx->stream) != hipSuccess)
        return SVT_AMD_ERR_DEVICE;
    pic->written = true;
    return SVT_AMD_OK;
}
void svt_amd_md_state_free(SvtAmdEncDecPicture *pic); /* md_kernels.hip */
/* encdec_kernels.hip: the encode pass of every LCU of the picture from work records the mode-decision kernel left in HBM, queued on ctx's stream behind it */
int svt_amd_ep_launch_behind_md(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const void *d_works, void *d_results, int n_active, const unsigned *d_order, int inter, int tiles);

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

/* development aid (builds with -DMD_TRACE only; tools/md_trace.py): time stamps of lane 0 of every wave at marks along the mode decision's unit chain, for ONE chosen
 * LCU and two consecutive units - a poor man's thread trace that shows which wave the chain is waiting for at every barrier */
#ifdef MD_TRACE
#define MD_TRACE_N 120
static __shared__ unsigned long long g_md_trace[4][MD_TRACE_N];
static __shared__ int g_md_trace_n[4];
static __shared__ int g_md_trace_on;
#define MD_TR(id)                                                                                                                   \
    do {                                                                                                                            \
        if (g_md_trace_on && (threadIdx.x & 63) == 0) {                                                                             \
            const int w_ = threadIdx.x >> 6, n_ = g_md_trace_n[w_];                                                                 \
            if (n_ < MD_TRACE_N) {                                                                                                  \
                g_md_trace[w_][n_] = ((unsigned long long)(id) << 48) | (__builtin_readcyclecounter() & 0xFFFFFFFFFFFFull);         \
                g_md_trace_n[w_] = n_ + 1;                                                                                          \
            }                                                                                                                       \
        }                                                                                                                           \
    } while (0)
#else
#define MD_TR(id) do { } while (0)
#endif

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
    static constexpr int WP = 40, TP = 44;
    alignas(16) T win[39 * WP];        /* reference window: (32 + 7) rows x columns, rows 8-byte chunks */
    alignas(16) int16_t tmp[32 * TP];  /* horizontally filtered rows, TRANSPOSED: column x at tmp + x * TP (a lane of the vertical pass reads one run of it) */
    int16_t raw[32 * 32];              /* list-0 intermediate of a bi-predicted tile */
};

/* The LUMA reference samples an LCU's motion-compensated candidates are likely to read, staged ONCE per LCU and list: the (64 + 2 R + 7)^2 samples of the padded reference
 * plane around the LCU displaced by a centre vector (the mode decision: the 64x64 unit's motion-estimation vector).  A candidate whose window lies inside takes it from
 * here (an LDS copy, ~0.3 K clocks) instead of from HBM (a round trip of ~5 K clocks on the unit chain, two for a bi-predicted candidate); any other candidate reads
 * global memory as before.  Filled by ep_ref_windows_fill with the core's own clamped addressing, so the samples are the ones the core would have loaded. */
#ifdef EP_DEBUG_WINDOW_COUNTS
static __device__ unsigned g_ep_dbg_counts[8]; /* development aid: lookups with a valid window / an invalid one / none, and hits; [4..7]: clocks / 16 of set-up, window, horizontal, vertical */
#define EP_DBG_CLK(k)                                                                              \
    do {                                                                                           \
        if (lane == 0 && !chroma) {                                                                \
            const unsigned long long c_ = __builtin_readcyclecounter();                            \
            atomicAdd(&g_ep_dbg_counts[k], (unsigned)((c_ - dbg_t_) >> 4));                         \
            dbg_t_ = c_;                                                                           \
        }                                                                                          \
    } while (0)
#else
#define EP_DBG_CLK(k)
#endif
struct EpRefWindows {
    static constexpr int R = 8, W = 64 + 2 * R + 7, P = 88, H = W;
    int x0[2], y0[2]; /* padded-plane coordinates of the window's first sample, per list */
    int valid[2];
    alignas(16) uint8_t pix[2][H * P + 16];
};
/* ... and the CHROMA samples around the same position, both planes (the mode decision of CHROMA_MODE_FULL LCUs predicts the chroma pair of every candidate in both of its
 * loops: a window from HBM is a round trip of ~2 K clocks per list and plane on the unit chain).  Rows of words (P a multiple of 4): the filter reads them where they lie. */
struct EpRefWindowsC {
    static constexpr int R = 4, W = 32 + 2 * R + 3, P = 44, H = W;
    int x0[2], y0[2]; /* padded-plane coordinates (chroma samples) of the window's first sample, per list */
    int valid[2];
    alignas(16) uint8_t pix[2][2][H * P + 20]; /* [list][Cb, Cr] */
};
/* by all 256 threads of the workgroup; mv[l] = the centre vector of list l in quarter samples */
__device__ __forceinline__ void ep_ref_windows_fill(const EpRefPlanes *refs, int lcu_x, int lcu_y, const bool use[2], const int16_t (*mv)[2], EpRefWindows &RW, int t);

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
/* the core: the 2Nx2N unit at luma position (abs_x, abs_y) of the picture, size N, direction / vectors as given; store(x, y) = where sample
 * (x, y) of the unit's plane-p block goes */
template <typename T, typename Store>
__device__ __forceinline__ void ep_inter_predict_core(const EpPicture &P, int abs_x, int abs_y, int N, int inter_dir, const int16_t (*mv)[2], int p, int lane,
                                                      EpMcScratch<T> &M, Store store, int tile_first = 0, int tile_step = 1, const EpRefWindows *RW = nullptr)
{
    constexpr int WP = EpMcScratch<T>::WP, TP = EpMcScratch<T>::TP;
    constexpr int s1 = sizeof(T) == 1 ? 0 : 2, maxv = sizeof(T) == 1 ? 255 : 1023;
    const bool chroma = p != 0;
    const int B = (sizeof(T) == 2 || !chroma) ? 8192 : 0;
    const int n = chroma ? N >> 1 : N, tn = n > 32 ? 32 : n, lgt = 31 - __clz(tn);
    const int ntaps = chroma ? 4 : 8, first = chroma ? -1 : -3, rows = tn + ntaps - 1;
    const bool bi = inter_dir == 2;
#ifdef EP_DEBUG_WINDOW_COUNTS
    unsigned long long dbg_t_ = __builtin_readcyclecounter();
#endif
    /* the block goes in 32x32 tiles (one for blocks up to 32x32); tile_first / tile_step let several waves share the tiles of a 64x64 block */
    const int ntile = n > 32 ? 4 : 1;
    for (int ti = tile_first; ti < ntile; ti += tile_step) {
        {
            const int ty0 = (ti >> 1) << 5, tx0 = (ti & 1) << 5;
            bool second = false;
            for (int l = 0; l < 2; l++) {
                if (!(bi || inter_dir == l))
                    continue;
                const EpRefPlanes &R = P.ref[l];
                const int qx = min(max(((abs_x + R.originX) << 2) + mv[l][0], (R.originX - 71) << 2), (R.width + R.originX + 7) << 2);
                const int qy = min(max(((abs_y + R.originY) << 2) + mv[l][1], (R.originY - 71) << 2), (R.height + R.originY + 7) << 2);
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
                /* window: 8-byte chunks of the rows, a lane per chunk, every load issued before the first store (one memory latency per window, not one per row).
                 * The chunks are not aligned (global memory takes that); one that leaves the plane is read sample by sample from clamped addresses. */
                EP_DBG_CLK(4);
                bool staged = false;
#ifdef EP_DEBUG_WINDOW_COUNTS
                if (lane == 0 && !chroma)
                    atomicAdd(&g_ep_dbg_counts[RW ? (RW->valid[l] ? 0 : 1) : 2], 1u);
#endif
                if constexpr (sizeof(T) == 1) {
                    /* the LCU's staged reference samples (EpRefWindows), when the tile's window lies inside: chunks of 8 bytes at any byte offset - three aligned words
                     * and two v_alignbyte each */
                    if (RW && !chroma && RW->valid[l]) {
                        const int cpr = (rows + 7) >> 3, rx = ix + first - RW->x0[l], ry = iy + first - RW->y0[l];
                        if (rx >= 0 && ry >= 0 && rx + cpr * 8 <= EpRefWindows::P && ry + rows <= EpRefWindows::H) {
                            staged = true;
#ifdef EP_DEBUG_WINDOW_COUNTS
                            if (lane == 0)
                                atomicAdd(&g_ep_dbg_counts[3], 1u);
#endif
                            const int nchunk = rows * cpr, inv = (65536 + cpr - 1) / cpr;
                            for (int i = lane; i < nchunk; i += 64) {
                                const int j = (i * inv) >> 16, m = i - j * cpr, o = (ry + j) * EpRefWindows::P + rx + m * 8, sh = o & 3;
                                const uint32_t *wp = reinterpret_cast<const uint32_t *>(&RW->pix[l][o & ~3]);
                                const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2];
                                uint2 v;
                                v.x = __builtin_amdgcn_alignbyte(w1, w0, sh), v.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
                                *reinterpret_cast<uint2 *>(&M.win[j * WP + m * 8]) = v;
                            }
                        }
                    }
                }
                if (!staged) {
                    constexpr int SPC = 8 / (int)sizeof(T); /* samples per chunk */
                    const int cpr = (rows + SPC - 1) / SPC, nchunk = rows * cpr, inv = (65536 + cpr - 1) / cpr;
                    const int base0 = (iy + first) * stride + ix + first;
                    for (int i0 = 0; i0 < nchunk; i0 += 256) {
                        uint2 v[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int i = i0 + u * 64 + lane;
                            if (i < nchunk) {
                                const int j = (i * inv) >> 16, m = i - j * cpr, idx = base0 + j * stride + m * SPC;
                                if (idx >= 0 && idx + SPC <= last + 1) {
                                    __builtin_memcpy(&v[u], plane + idx, 8);
                                } else {
                                    T e[SPC];
#pragma unroll
                                    for (int q = 0; q < SPC; q++)
                                        e[q] = plane[min(max(idx + q, 0), last)];
                                    __builtin_memcpy(&v[u], e, 8);
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int i = i0 + u * 64 + lane;
                            if (i < nchunk) {
                                const int j = (i * inv) >> 16, m = i - j * cpr;
                                *reinterpret_cast<uint2 *>(&M.win[j * WP + m * SPC]) = v[u];
                            }
                        }
                    }
                }
                EP_WAVE_SYNC();
                EP_DBG_CLK(5);
                /* horizontal pass of every window row: a lane slides the taps over a run of seg outputs (seg + taps - 1 samples, read as whole words) */
                const int seg = tn < 8 ? tn : 8, lgs = tn < 8 ? lgt : 3, spr = tn >> lgs; /* runs per row */
                for (int i = lane; i < rows * spr; i += 64) {
                    const int j = i >> (lgt - lgs), x0 = (i & (spr - 1)) << lgs;
                    constexpr int NW = 4 * (int)sizeof(T); /* words holding 16 samples */
                    uint32_t w[NW];
                    const uint32_t *wp = reinterpret_cast<const uint32_t *>(&M.win[j * WP + x0]);
#pragma unroll
                    for (int k = 0; k < NW; k++)
                        w[k] = wp[k];
                    int in[15];
#pragma unroll
                    for (int k = 0; k < 15; k++)
                        in[k] = sizeof(T) == 1 ? (int)((w[k >> 2] >> (8 * (k & 3))) & 0xFFu) : (int)((w[(k >> 1) % NW] >> (16 * (k & 1))) & 0xFFFFu);
#pragma unroll
                    for (int o = 0; o < 8; o++)
                        if (o < seg) {
                            int hs = 0;
#pragma unroll
                            for (int k = 0; k < 8; k++)
                                if (k < ntaps)
                                    hs += tx[k] * in[o + k];
                            M.tmp[(x0 + o) * TP + j] = (int16_t)((hs - (B << s1)) >> s1);
                        }
                }
                EP_WAVE_SYNC();
                EP_DBG_CLK(6);
                /* vertical pass: a lane owns a column and a run of rpl rows (rpl + taps - 1 reads: one stretch of the transposed rows) */
                const int rpl = tn >= 8 ? (tn * tn) >> 6 : 1, run = rpl < 1 ? 1 : rpl; /* 32: 16, 16: 4, 8: 1, 4: 1 */
                {
                    const int x = lane & (tn - 1), y0 = (lane >> lgt) * run;
                    if (y0 < tn) {
                        int in[23];
                        const int16_t *col = &M.tmp[x * TP + y0];
                        if (run >= 4) { /* y0 is a multiple of 4 and TP of 4: 8-byte words */
                            const uint2 *cp = reinterpret_cast<const uint2 *>(col);
#pragma unroll
                            for (int k = 0; k < 6; k++)
                                if (4 * k < run + ntaps - 1) {
                                    const uint2 d = cp[k];
                                    in[4 * k] = (int)(int16_t)(d.x & 0xFFFFu), in[4 * k + 1] = (int)(int16_t)(d.x >> 16);
                                    in[4 * k + 2] = (int)(int16_t)(d.y & 0xFFFFu);
                                    if (4 * k + 3 < 23)
                                        in[4 * k + 3] = (int)(int16_t)(d.y >> 16);
                                } else {
                                    in[4 * k] = in[4 * k + 1] = in[4 * k + 2] = 0;
                                    if (4 * k + 3 < 23)
                                        in[4 * k + 3] = 0;
                                }
                        } else {
#pragma unroll
                            for (int k = 0; k < 23; k++)
                                in[k] = k < run + ntaps - 1 ? (int)col[k] : 0;
                        }
#pragma unroll
                        for (int o = 0; o < 16; o++)
                            if (o < run) {
                                int sum = 0;
#pragma unroll
                                for (int k = 0; k < 8; k++)
                                    if (k < ntaps)
                                        sum += tv[k] * in[o + k];
                                const int y = y0 + o, i = (y << lgt) + x;
                                T *dst = store(tx0 + x, ty0 + y);
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
                EP_DBG_CLK(7);
                second = true;
            }
        }
    }
}

/* ---- the same interpolation for 8-bit planes with everything a wave repeats per call decided at compile time (tile size, tap count) and the multiply-adds on the dot-product
 * units: v_dot4c_i32_i8 for the horizontal pass - the window's samples as SIGNED bytes s - 128 (one v_xor per word), so that sum t_k s_k - 8192 = sum t_k (s_k - 128): exactly the
 * offset the reference subtracts from the luma intermediate (the taps of a position sum to 64) -, v_dot2c_i32_i16 for the vertical pass on the transposed intermediate's
 * packed pairs.  Same values as ep_inter_predict_core<uint8_t> sample for sample (the mode decision's tests compare decisions that hinge on them); 2 - 3x fewer instructions
 * on the unit chain of the mode decision, where a wave has nobody to hide them behind. */
constexpr uint32_t ep_pack4(int a, int b, int c, int d) { return (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(c & 0xFF) << 16) | ((uint32_t)(d & 0xFF) << 24); }
constexpr uint32_t ep_pack2(int a, int b) { return (uint32_t)(a & 0xFFFF) | ((uint32_t)(b & 0xFFFF) << 16); }
/* taps of fractional position f (wave-uniform) as SELECTS BETWEEN IMMEDIATES - a table, in whatever memory, is a load (and its latency) on every call: bytes k = 0..3 / 4..7 for
 * the horizontal pass (h), pairs of 16-bit values for the vertical one (v).  Luma: the 8-tap filters of quarter positions 0..3; chroma: the 4-tap filters of eighth positions 0..7
 * (H.265 tables 8-11 / 8-12 = c_ep_luma_taps / c_ep_chroma_taps above). */
#define EP_SEL4(f, a, b, c, d) ((f) == 0 ? (a) : (f) == 1 ? (b) : (f) == 2 ? (c) : (d))
#define EP_SEL8(f, a, b, c, d, e, g, h, i) ((f) < 4 ? EP_SEL4(f, a, b, c, d) : EP_SEL4((f) - 4, e, g, h, i))
template <bool CHROMA>
__device__ __forceinline__ void ep_taps8(int f, uint32_t h[2], uint32_t v[4])
{
    if (!CHROMA) {
        h[0] = EP_SEL4(f, ep_pack4(0, 0, 0, 64), ep_pack4(-1, 4, -10, 58), ep_pack4(-1, 4, -11, 40), ep_pack4(0, 1, -5, 17));
        h[1] = EP_SEL4(f, ep_pack4(0, 0, 0, 0), ep_pack4(17, -5, 1, 0), ep_pack4(40, -11, 4, -1), ep_pack4(58, -10, 4, -1));
        v[0] = EP_SEL4(f, ep_pack2(0, 0), ep_pack2(-1, 4), ep_pack2(-1, 4), ep_pack2(0, 1));
        v[1] = EP_SEL4(f, ep_pack2(0, 64), ep_pack2(-10, 58), ep_pack2(-11, 40), ep_pack2(-5, 17));
        v[2] = EP_SEL4(f, ep_pack2(0, 0), ep_pack2(17, -5), ep_pack2(40, -11), ep_pack2(58, -10));
        v[3] = EP_SEL4(f, ep_pack2(0, 0), ep_pack2(1, 0), ep_pack2(4, -1), ep_pack2(4, -1));
    } else {
        h[0] = EP_SEL8(f, ep_pack4(0, 64, 0, 0), ep_pack4(-2, 58, 10, -2), ep_pack4(-4, 54, 16, -2), ep_pack4(-6, 46, 28, -4), ep_pack4(-4, 36, 36, -4), ep_pack4(-4, 28, 46, -6),
                       ep_pack4(-2, 16, 54, -4), ep_pack4(-2, 10, 58, -2));
        h[1] = 0;
        v[0] = EP_SEL8(f, ep_pack2(0, 64), ep_pack2(-2, 58), ep_pack2(-4, 54), ep_pack2(-6, 46), ep_pack2(-4, 36), ep_pack2(-4, 28), ep_pack2(-2, 16), ep_pack2(-2, 10));
        v[1] = EP_SEL8(f, ep_pack2(0, 0), ep_pack2(10, -2), ep_pack2(16, -2), ep_pack2(28, -4), ep_pack2(36, -4), ep_pack2(46, -6), ep_pack2(54, -4), ep_pack2(58, -2));
        v[2] = v[3] = 0;
    }
}
__device__ __forceinline__ int ep_sdot2(uint32_t a, uint32_t b, int c)
{
    typedef short v2s __attribute__((ext_vector_type(2)));
    v2s x, y;
    __builtin_memcpy(&x, &a, 4), __builtin_memcpy(&y, &b, 4);
    return __builtin_amdgcn_sdot2(x, y, c, false);
}
/* both passes of one tile of TN x TN samples of one list: window in M.win -> dst / M.raw.  mode: 0 uni-prediction, 1 first list of a bi-predicted tile, 2 its second list */
/* DIRECT: the window is read where it lies - dwin + doff, rows dpitch bytes apart (dwin 4-byte aligned, dpitch a multiple of 4: the byte misalignment is the same for every
 * row and segment, one v_alignbyte per word) - instead of from an 8-byte-aligned copy in M.win: a staged reference window (EpRefWindows) needs no copy pass */
template <int TN, bool CHROMA, bool DIRECT = false>
__device__ __forceinline__ void ep_mc8_tile(EpMcScratch<uint8_t> &M, int lane, int fx, int fy, int mode, uint8_t *dst, int pitch, const uint8_t *dwin = nullptr, int doff = 0,
                                            int dpitch = 0)
{
    constexpr int WP = EpMcScratch<uint8_t>::WP, TP = EpMcScratch<uint8_t>::TP;
    constexpr int NT = CHROMA ? 4 : 8, ROWS = TN + NT - 1, SEG = TN < 8 ? TN : 8, SPR = TN / SEG, ITEMS = ROWS * SPR;
    constexpr int LGT = TN == 32 ? 5 : TN == 16 ? 4 : TN == 8 ? 3 : 2;
    uint32_t hx[2], vx[4], hy[2], vy[4];
    ep_taps8<CHROMA>(fx, hx, vx);
    ep_taps8<CHROMA>(fy, hy, vy);
    /* horizontal pass: an item = a run of SEG outputs of one window row, 16 bytes of the row as four words */
#pragma unroll
    for (int i0 = 0; i0 < ITEMS; i0 += 64) {
        const int i = i0 + lane;
        if (i < ITEMS) {
            const int j = SPR == 1 ? i : i / SPR, x0 = SPR == 1 ? 0 : (i - j * SPR) * SEG;
            uint32_t w[4];
            if (DIRECT) {
                const int sh = __builtin_amdgcn_readfirstlane(doff & 3);
                const uint32_t *wp = reinterpret_cast<const uint32_t *>(dwin + (doff & ~3) + j * dpitch + x0);
                uint32_t r[5];
#pragma unroll
                for (int k = 0; k < 5; k++)
                    r[k] = (k - 1) * 4 < SEG + NT - 1 ? wp[k] : 0u;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    w[k] = (k * 4 < SEG + NT - 1 ? __builtin_amdgcn_alignbyte(r[k + 1], r[k], sh) : 0u) ^ 0x80808080u;
            } else {
                const uint32_t *wp = reinterpret_cast<const uint32_t *>(&M.win[j * WP + x0]);
#pragma unroll
                for (int k = 0; k < 4; k++)
                    w[k] = (k * 4 < SEG + NT - 1 ? wp[k] : 0u) ^ 0x80808080u;
            }
#pragma unroll
            for (int o = 0; o < SEG; o++) {
                const uint32_t lo = (o & 3) ? __builtin_amdgcn_alignbyte(w[(o >> 2) + 1], w[o >> 2], o & 3) : w[o >> 2];
                int hs;
                if (CHROMA) {
                    hs = __builtin_amdgcn_sdot4((int)hx[0], (int)lo, 8192, false); /* no offset is subtracted from an 8-bit chroma intermediate: + 128 * 64 */
                } else {
                    const uint32_t hi = (o & 3) ? __builtin_amdgcn_alignbyte(w[((o >> 2) + 2) & 3], w[(o >> 2) + 1], o & 3) : w[(o >> 2) + 1];
                    hs = __builtin_amdgcn_sdot4((int)hx[1], (int)hi, __builtin_amdgcn_sdot4((int)hx[0], (int)lo, 0, false), false);
                }
                M.tmp[(x0 + o) * TP + j] = (int16_t)hs;
            }
        }
    }
    EP_WAVE_SYNC();
    MD_TR(43);
    /* vertical pass: a lane owns a column and RUN rows; the column of the transposed intermediate as packed pairs */
    constexpr int RUN = TN >= 8 ? (TN * TN) / 64 : 1, NIN = RUN + NT - 1, NW = (NIN + 2) / 2 + 1;
    const int x = lane & (TN - 1), y0 = (lane >> LGT) * RUN;
    if (y0 < TN) {
        const int base = y0 & ~1, odd = (y0 & 1) * 16; /* RUN >= 4: y0 is a multiple of 4 */
        const uint32_t *cp = reinterpret_cast<const uint32_t *>(&M.tmp[x * TP + base]);
        uint32_t d[NW];
#pragma unroll
        for (int k = 0; k < NW; k++)
            d[k] = cp[k];
#pragma unroll
        for (int o = 0; o < RUN; o++) {
            int sum = 0;
#pragma unroll
            for (int k = 0; k < NT / 2; k++) {
                const int m = (o >> 1) + k;
                const uint32_t pr = RUN >= 4 ? ((o & 1) ? __builtin_amdgcn_alignbit(d[m + 1], d[m], 16) : d[m]) : __builtin_amdgcn_alignbit(d[m + 1], d[m], odd);
                sum = ep_sdot2(vy[k], pr, sum);
            }
            const int y = y0 + o, idx = (y << LGT) + x;
            if (mode == 0) {
                dst[y * pitch + x] = (uint8_t)min(255, max(0, (sum + ((CHROMA ? 0 : 8192) << 6) + (1 << 11)) >> 12));
            } else if (mode == 1) {
                M.raw[idx] = (int16_t)(sum >> 6);
            } else { /* BiPredClipping (Offset5 / ChromaOffset5, Codec/EbDefinitions.h:1022-1030) */
                const int a = (int)M.raw[idx] + (int)(int16_t)(sum >> 6);
                dst[y * pitch + x] = (uint8_t)min(255, max(0, (a + (CHROMA ? 64 : 16448)) >> 7));
            }
        }
    }
    EP_WAVE_SYNC();
}
/* the core for 8-bit planes: dst = the block's first sample, pitch in samples (plane p of the 2Nx2N unit at (abs_x, abs_y)) */
__device__ __forceinline__ void ep_inter_predict_core8(const EpRefPlanes *refs /* [2]: P.ref, or the caller's copy of it in LDS */, int abs_x, int abs_y, int N, int inter_dir,
                                                       const int16_t (*mv)[2], int p, int lane, EpMcScratch<uint8_t> &M, uint8_t *dst, int pitch, int tile_first, int tile_step,
                                                       const EpRefWindows *RW)
{
    constexpr int WP = EpMcScratch<uint8_t>::WP;
    const bool chroma = p != 0;
    const int n = chroma ? N >> 1 : N, tn = n > 32 ? 32 : n;
    const int ntaps = chroma ? 4 : 8, first = chroma ? -1 : -3, rows = tn + ntaps - 1;
    const bool bi = inter_dir == 2;
    const int ntile = n > 32 ? 4 : 1;
#ifdef EP_DEBUG_WINDOW_COUNTS
    unsigned long long dbg_t_ = __builtin_readcyclecounter();
#endif
    for (int ti = tile_first; ti < ntile; ti += tile_step) {
        const int ty0 = (ti >> 1) << 5, tx0 = (ti & 1) << 5;
        bool second = false;
        for (int l = 0; l < 2; l++) {
            if (!(bi || inter_dir == l))
                continue;
            MD_TR(40);
            const EpRefPlanes R = refs[l]; /* one read of the whole record */
            const int qx = min(max(((abs_x + R.originX) << 2) + mv[l][0], (R.originX - 71) << 2), (R.width + R.originX + 7) << 2);
            const int qy = min(max(((abs_y + R.originY) << 2) + mv[l][1], (R.originY - 71) << 2), (R.height + R.originY + 7) << 2);
            const int ix = (chroma ? qx >> 3 : qx >> 2) + tx0, iy = (chroma ? qy >> 3 : qy >> 2) + ty0;
            const int fx = __builtin_amdgcn_readfirstlane(chroma ? qx & 7 : qx & 3), fy = __builtin_amdgcn_readfirstlane(chroma ? qy & 7 : qy & 3);
            const int cpr = (rows + 7) >> 3, nchunk = rows * cpr, inv = (65536 + cpr - 1) / cpr;
            bool staged = false;
            MD_TR(41);
            EP_DBG_CLK(4);
#ifdef EP_DEBUG_WINDOW_COUNTS
            if (lane == 0 && !chroma)
                atomicAdd(&g_ep_dbg_counts[0], 1u);
#endif
            if (RW && !chroma && RW->valid[l]) {
                const int rx = ix + first - RW->x0[l], ry = iy + first - RW->y0[l];
                if (rx >= 0 && ry >= 0 && rx + cpr * 8 <= EpRefWindows::P && ry + rows <= EpRefWindows::H) {
                    staged = true;
                    for (int i = lane; i < nchunk; i += 64) {
                        const int j = (i * inv) >> 16, m = i - j * cpr, o = (ry + j) * EpRefWindows::P + rx + m * 8, sh = o & 3;
                        const uint32_t *wp = reinterpret_cast<const uint32_t *>(&RW->pix[l][o & ~3]);
                        const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2];
                        uint2 v;
                        v.x = __builtin_amdgcn_alignbyte(w1, w0, sh), v.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
                        *reinterpret_cast<uint2 *>(&M.win[j * WP + m * 8]) = v;
                    }
                }
            }
            if (!staged) {
                const int stride = (int)R.stride[chroma], last = R.size[chroma] - 1;
                const uint8_t *plane = (const uint8_t *)R.plane[p];
                const int base0 = (iy + first) * stride + ix + first;
                for (int i0 = 0; i0 < nchunk; i0 += 256) {
                    uint2 v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int i = i0 + u * 64 + lane;
                        if (i < nchunk) {
                            const int j = (i * inv) >> 16, m = i - j * cpr, idx = base0 + j * stride + m * 8;
                            if (idx >= 0 && idx + 8 <= last + 1) {
                                __builtin_memcpy(&v[u], plane + idx, 8);
                            } else {
                                uint8_t e[8];
#pragma unroll
                                for (int q = 0; q < 8; q++)
                                    e[q] = plane[min(max(idx + q, 0), last)];
                                __builtin_memcpy(&v[u], e, 8);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int i = i0 + u * 64 + lane;
                        if (i < nchunk) {
                            const int j = (i * inv) >> 16, m = i - j * cpr;
                            *reinterpret_cast<uint2 *>(&M.win[j * WP + m * 8]) = v[u];
                        }
                    }
                }
            }
            EP_WAVE_SYNC();
            MD_TR(staged ? 42 : 46);
            EP_DBG_CLK(5);
            const int mode = !bi ? 0 : (second ? 2 : 1);
            uint8_t *td = dst + ty0 * pitch + tx0;
            if (!chroma) {
                switch (tn) {
                case 32: ep_mc8_tile<32, false>(M, lane, fx, fy, mode, td, pitch); break;
                case 16: ep_mc8_tile<16, false>(M, lane, fx, fy, mode, td, pitch); break;
                default: ep_mc8_tile<8, false>(M, lane, fx, fy, mode, td, pitch); break;
                }
            } else {
                switch (tn) {
                case 32: ep_mc8_tile<32, true>(M, lane, fx, fy, mode, td, pitch); break;
                case 16: ep_mc8_tile<16, true>(M, lane, fx, fy, mode, td, pitch); break;
                case 8: ep_mc8_tile<8, true>(M, lane, fx, fy, mode, td, pitch); break;
                default: ep_mc8_tile<4, true>(M, lane, fx, fy, mode, td, pitch); break;
                }
            }
            MD_TR(44);
            EP_DBG_CLK(6);
            second = true;
        }
    }
}

__device__ __forceinline__ void ep_ref_windows_fill(const EpRefPlanes *refs, int lcu_x, int lcu_y, const bool use[2], const int16_t (*mv)[2], EpRefWindows &RW, int t)
{
    for (int l = 0; l < 2; l++) {
        if (!use[l]) {
            if (t == 0)
                RW.valid[l] = 0;
            continue;
        }
        const EpRefPlanes &R = refs[l];
        /* the position clamp of the core (Codec/EbInterPrediction.c:802-812) applied to the LCU's origin displaced by the centre vector */
        const int qx = min(max(((lcu_x + R.originX) << 2) + mv[l][0], (R.originX - 71) << 2), (R.width + R.originX + 7) << 2);
        const int qy = min(max(((lcu_y + R.originY) << 2) + mv[l][1], (R.originY - 71) << 2), (R.height + R.originY + 7) << 2);
        const int x0 = (qx >> 2) - 3 - EpRefWindows::R, y0 = (qy >> 2) - 3 - EpRefWindows::R;
        const int stride = (int)R.stride[0], last = R.size[0] - 1;
        const uint8_t *plane = (const uint8_t *)R.plane[0];
        constexpr int CPR = EpRefWindows::P / 8, NCH = EpRefWindows::H * CPR, NU = (NCH + 255) / 256;
        /* every load of a thread issued before its first store: ONE memory latency per window (the mode decision re-stages a window behind its wait for the neighbours,
         * on the picture's critical path) */
        uint2 v[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int i = t + 256 * u;
            v[u] = make_uint2(0u, 0u);
            if (i < NCH) {
                const int j = i / CPR, m = i - j * CPR, idx = (y0 + j) * stride + x0 + m * 8;
                if (idx >= 0 && idx + 8 <= last + 1) {
                    __builtin_memcpy(&v[u], plane + idx, 8);
                } else {
                    uint32_t lo = 0, hi = 0;
                    for (int q = 0; q < 4; q++)
                        lo |= (uint32_t)plane[min(max(idx + q, 0), last)] << (8 * q), hi |= (uint32_t)plane[min(max(idx + 4 + q, 0), last)] << (8 * q);
                    v[u] = make_uint2(lo, hi);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int i = t + 256 * u;
            if (i < NCH) {
                const int j = i / CPR, m = i - j * CPR;
                *reinterpret_cast<uint2 *>(&RW.pix[l][j * EpRefWindows::P + m * 8]) = v[u];
            }
        }
        if (t == 0)
            RW.x0[l] = x0, RW.y0[l] = y0, RW.valid[l] = 1;
    }
}

__device__ __forceinline__ void ep_ref_windows_fill_chroma(const EpRefPlanes *refs, int lcu_x, int lcu_y, const bool use[2], const int16_t (*mv)[2], EpRefWindowsC &RW, int t)
{
    for (int l = 0; l < 2; l++) {
        if (!use[l]) {
            if (t == 0)
                RW.valid[l] = 0;
            continue;
        }
        const EpRefPlanes &R = refs[l];
        const int qx = min(max(((lcu_x + R.originX) << 2) + mv[l][0], (R.originX - 71) << 2), (R.width + R.originX + 7) << 2);
        const int qy = min(max(((lcu_y + R.originY) << 2) + mv[l][1], (R.originY - 71) << 2), (R.height + R.originY + 7) << 2);
        const int x0 = (qx >> 3) - 1 - EpRefWindowsC::R, y0 = (qy >> 3) - 1 - EpRefWindowsC::R;
        const int stride = (int)R.stride[1], last = R.size[1] - 1;
        constexpr int WPR = EpRefWindowsC::P / 4, NW = EpRefWindowsC::H * WPR;
        for (int i = t; i < 2 * NW; i += 256) {
            const int pl = i >= NW, r = i - pl * NW, j = r / WPR, m = r - j * WPR, idx = (y0 + j) * stride + x0 + m * 4;
            const uint8_t *plane = (const uint8_t *)R.plane[1 + pl];
            uint32_t v;
            if (idx >= 0 && idx + 4 <= last + 1) {
                __builtin_memcpy(&v, plane + idx, 4);
            } else { /* the core's clamped addressing */
                uint8_t e[4];
#pragma unroll
                for (int q = 0; q < 4; q++)
                    e[q] = plane[min(max(idx + q, 0), last)];
                __builtin_memcpy(&v, e, 4);
            }
            *reinterpret_cast<uint32_t *>(&RW.pix[l][pl][j * EpRefWindowsC::P + m * 4]) = v;
        }
        if (t == 0)
            RW.x0[l] = x0, RW.y0[l] = y0, RW.valid[l] = 1;
    }
}

template <typename T>
__device__ __forceinline__ void ep_inter_predict_plane(const EpPicture &P, EpLocal<T> &L, const EpFlags &F, const LcuCu &cu, int p, int lane, EpMcScratch<T> &M)
{
    const int lx = p ? cu.x >> 1 : cu.x, ly = p ? cu.y >> 1 : cu.y;
    ep_inter_predict_core<T>(P, F.lcu_x + cu.x, F.lcu_y + cu.y, cu.size, cu.inter_dir, cu.mv, p, lane, M, [&](int x, int y) { return L.at(p, lx + x, ly + y); });
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
    int16_t tiles[4][2 * TxRegTile<32>::UNIT]; /* 64 / N units of TxRegTile<N>::UNIT each fit for every N; one per wave */
    EpMcScratch<T> mc[4];                      /* inter units: one per wave (the predictions of an LCU's inter units are made by all four waves) */
    unsigned next_task;                        /* the waves' draw from the LCU's list of (inter unit, plane) tasks */
    unsigned decide_lock;                      /* qbuf / cfbuf / Pq below exist once: luma planes that decide their cbf or re-decide levels take turns */
    int16_t qbuf[32 * 32];                     /* luma cbf decision of AMVP units; levels of the PM-core re-decision */
    int16_t cfbuf[32 * 32];                    /* PM-core: the luma unit's coefficients */
    int16_t Pq[64][16];                        /* PM-core: per-lane scratch of the 4x4 rate estimate */
};

/* the coding-unit loop of one LCU, by one workgroup of 256 threads */
/* one lane: wait until the LCUs this LCU's units read from are finished (see k_encode_picture), then acquire */
template <typename WorkT>
__device__ __forceinline__ void ep_wait_neighbours(const WorkT &W, int lcu, int wl, const unsigned *done, unsigned epoch)
{
    bool intra = false;
    for (int i = 0; i < W.num_cus; i++)
        intra |= W.cu[i].pred_mode == 2;
    if (intra) {
        const int x = W.lcu_x >> 6;
        const bool right_edge = W.tile_right || x + 1 >= wl;
        const int dep[4] = {W.tile_left ? -1 : lcu - 1, W.tile_top ? -1 : lcu - wl, (W.tile_top || W.tile_left) ? -1 : lcu - wl - 1,
                            (W.tile_top || right_edge) ? -1 : lcu - wl + 1};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (dep[k] >= 0)
                while (__hip_atomic_load(&done[dep[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch)
                    __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); /* this CU's / XCD's caches drop what they held of the neighbours' samples */
}

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
    if (t == 0)
        S.next_task = 0, S.decide_lock = 0;
    __syncthreads();
    /* ---- the predictions of the LCU's inter units first, by all four waves (round 3): an inter unit predicts from the reference pictures alone, so its three
     * planes are tasks nothing in the LCU has to wait for; the waves draw them - luma planes first, they are the long ones - and leave the predicted samples at
     * the units' positions.  The unit loop below then only transforms them, in order (an intra unit in between reads the reconstruction of the units before it). ---- */
    {
        const int ntasks = 3 * num_cus;
        for (;;) {
            int k = 0;
            if (lane == 0)
                k = (int)atomicAdd(&S.next_task, 1u);
            k = __shfl(k, 0);
            if (k >= ntasks)
                break;
            const int ci = k < num_cus ? k : (k - num_cus) >> 1, p = k < num_cus ? 0 : 1 + ((k - num_cus) & 1);
            const LcuCu cu = L.cus[ci];
            if (cu.pred_mode != 1)
                continue;
            ep_inter_predict_plane<T>(P, L, F, cu, p, lane, S.mc[wave]);
            /* ... and its transform units (EbCodingLoop.c:3817-4400): residual of the prediction just made, nothing else of the LCU */
            const int N = cu.size;
            const int ntu = N == 64 ? 4 : 1, TS = N == 64 ? 32 : N, n = p ? TS >> 1 : TS;
            const bool amvp = cu.inter_kind == SVT_AMD_EP_INTER_AMVP;
            const EpDecide D = {P.cost, S.qbuf, F.full_lambda, N == TS ? F.cbf_bits[1] : F.cbf_bits[0], N == TS ? F.cbf_bits[3] : F.cbf_bits[2],
                                F.pm_core,      1,      S.cfbuf,       S.Pq};
            const bool shared_scratch = p == 0 && (amvp || F.pm_core) && cu.inter_kind != SVT_AMD_EP_INTER_SKIP;
            if (shared_scratch) {
                if (lane == 0)
                    while (atomicCAS(&S.decide_lock, 0u, 1u) != 0u)
                        __builtin_amdgcn_s_sleep(1);
                EP_WAVE_SYNC();
            }
            uint32_t any = 0;
            for (int tu = 0; tu < ntu; tu++) {
                const int tx = cu.x + ((tu & 1) << 5), ty = cu.y + ((tu >> 1) << 5);
                const int lx = p ? tx >> 1 : tx, ly = p ? ty >> 1 : ty;
                uint32_t o = 0;
                if (cu.inter_kind != SVT_AMD_EP_INTER_SKIP) {
                    const T *src = p == 0 ? L.src_y + ly * 64 + lx : L.src_c[p - 1] + ly * 32 + lx;
                    int16_t *coeff = p == 0 ? R.coeff_y + ly * 64 + lx : (p == 1 ? R.coeff_cb : R.coeff_cr) + ly * 32 + lx;
                    o = ep_encode_plane<T>(n, lane, src, p ? 32 : 64, L.at(p, lx, ly), (size_t)L.pitch(p), coeff, p ? 32 : 64, tiles[wave],
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
            if (shared_scratch) {
                EP_WAVE_SYNC();
                if (lane == 0)
                    atomicExch(&S.decide_lock, 0u);
            }
        }
    }
    __syncthreads();
    if (P.prof)
        c1 = __builtin_readcyclecounter(), c_pred += c1 - c0, c0 = c1;
    if (wave < 3) { /* wave p = plane p: its own pipeline over the unit list (luma is the long one) */
        const int p = wave;
        for (int ci = 0; ci < num_cus; ci++) {
            const LcuCu cu = L.cus[ci];
            const int N = cu.size;
            if (cu.pred_mode == 1) { /* INTER_MODE: predicted and transformed above; here the unit only takes its place in the coding order */
                if (P.prof)
                    c1 = __builtin_readcyclecounter();
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

#endif
