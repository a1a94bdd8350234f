/*
 * Open-loop intra search on the device (SURVEY.md 8a front half, row "OpenLoopIntraSearchLcu").
 *
 * One workgroup (4 wavefronts) per LCU.  Replaces, per LCU, the reference's serial chain
 *   UpdateNeighborSamplesArrayOpenLoop -> IntraPredictionOpenLoop -> NxMSadKernel -> candidate injection
 * (EbMotionEstimation.c:5053-5320, EbIntraPrediction.c:5222-5421) by:
 *   1. the source window rows/cols -1..95 of the LCU in LDS (covers the 2N left / 2N top neighbours of every CU),
 *   2. the reference-sample array of all 84 CUs at once (out-of-picture samples = 128, no smoothing),
 *   3. a task list (CU, mode): one wavefront per task, lanes over the CU's samples, prediction evaluated per
 *      sample in closed form (planar / DC+edge / V,H+edge / angular with on-the-fly projection), SAD by
 *      wave reduction,
 *   4. per-CU decision threads reproducing the injection tables; the one serial dependency of the reference
 *      (bestMode / stage1SadArray surviving from CU to CU when no mode beats 32*32*255) is detected and, only
 *      then, replayed serially by one thread.
 * Results use the SvtAmdOisLcuResult convention (include/svt_hevc_amd.h): written bitfields flagged.
 */
#include "svt_amd_internal.h"

#define WIN_W 100 /* row pitch in bytes: columns -4 .. 95 of the LCU (starts on a dword of the plane) */
#define WIN_X0 4  /* window column of LCU column 0 */
#define WIN_H 97
#define MAXK 35

struct OisShared {
    uint8_t win[WIN_H * WIN_W];         /* win[(y+1)*WIN_W + x + WIN_X0] = source sample (x,y) relative to the LCU */
    uint8_t refs[4 * 129 + 16 * 65 + 64 * 33 + 4]; /* per CU: left[0..2N-1] top-to-bottom, top-left, top[0..2N-1] */
    uint32_t sad[85][MAXK];
    uint32_t out_cand[85][SVT_AMD_OIS_MAX_CAND];
    uint8_t out_total[88];
    uint8_t dc[88];
    uint8_t nmodes[88];                 /* stage-1 modes to test per CU (P path) */
    int stale;
};

__device__ __forceinline__ void cu_geom(int cu, int &x, int &y, int &N, int &lg)
{
    if (cu < 5)
        N = 32, lg = 5, x = ((cu - 1) & 1) * 32, y = ((cu - 1) >> 1) * 32;
    else if (cu < 21)
        N = 16, lg = 4, x = ((cu - 5) & 3) * 16, y = ((cu - 5) >> 2) * 16;
    else
        N = 8, lg = 3, x = ((cu - 21) & 7) * 8, y = ((cu - 21) >> 3) * 8;
}
__device__ __forceinline__ int ref_base(int cu)
{
    return cu < 5 ? (cu - 1) * 129 : cu < 21 ? 516 + (cu - 5) * 65 : 516 + 1040 + (cu - 21) * 33;
}

__device__ __constant__ int8_t c_ang[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};
__device__ __constant__ int16_t c_inv[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256};
__device__ __constant__ uint8_t c_islice[7] = {0, 1, 10, 26, 2, 18, 34};
__device__ __constant__ uint8_t c_stage1[9] = {10, 26, 2, 18, 34, 6, 14, 22, 30};
__device__ __constant__ uint8_t c_inject[9][9] = {
    {10, 1, 0, 9, 11, 8, 12, 7, 13}, {26, 1, 0, 25, 27, 24, 28, 23, 29}, {2, 1, 0, 3, 4, 5, 7, 8, 9},
    {18, 1, 0, 17, 19, 16, 20, 15, 21}, {34, 1, 0, 33, 32, 29, 31, 27, 28}, {6, 1, 0, 7, 5, 4, 8, 3, 9},
    {14, 1, 0, 13, 15, 12, 16, 11, 17}, {22, 1, 0, 21, 23, 20, 24, 19, 25}, {30, 1, 0, 29, 31, 28, 32, 27, 33}};
__device__ __constant__ uint8_t c_isl_inject[5][3] = {{2, 4, 6}, {10, 6, 14}, {18, 14, 22}, {26, 22, 30}, {34, 32, 30}};
__device__ __constant__ int16_t c_ois_th[3][6][4] = {
    {{-20, 50, 150, 200}, {-20, 50, 150, 200}, {-20, 50, 100, 150}, {-20, 50, 200, 300}, {-20, 50, 200, 300}, {-20, 50, 200, 300}},
    {{-150, 0, 150, 200}, {-150, 0, 150, 200}, {-125, 0, 100, 150}, {-50, 50, 200, 300}, {-50, 50, 200, 300}, {-50, 50, 200, 300}},
    {{-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}}};

/* HEVC intra prediction of sample (x,y), unfiltered references, luma edge filters for N < 32.
 * r: left[0..2N-1], r[2N] = top-left, r[2N+1+j] = top[j]. */
__device__ __forceinline__ int predict_sample(int mode, int N, int lg, const uint8_t *r, int x, int y, int dc)
{
    const uint8_t *left = r, *top = r + 2 * N + 1;
    const int tl = r[2 * N];
    if (mode == 0)
        return ((N - 1 - x) * left[y] + (x + 1) * top[N] + (N - 1 - y) * top[x] + (y + 1) * left[N] + N) >> (lg + 1);
    if (mode == 1) {
        if (N < 32) {
            if (x == 0 && y == 0)
                return (left[0] + top[0] + 2 * dc + 2) >> 2;
            if (y == 0)
                return (top[x] + 3 * dc + 2) >> 2;
            if (x == 0)
                return (left[y] + 3 * dc + 2) >> 2;
        }
        return dc;
    }
    if (mode == 26) {
        if (N < 32 && x == 0)
            return min(255, max(0, top[0] + ((left[y] - tl) >> 1)));
        return top[x];
    }
    if (mode == 10) {
        if (N < 32 && y == 0)
            return min(255, max(0, left[0] + ((top[x] - tl) >> 1)));
        return left[y];
    }
    /* angular: main / side reference by direction */
    const bool vert = mode >= 18;
    const int d = vert ? mode - 26 : 10 - mode;      /* -8..8 */
    const int a = d < 0 ? -c_ang[-d] : c_ang[d];
    const int u = vert ? x : y, v = vert ? y : x;     /* u along the main reference, v across */
    const uint8_t *mainr = vert ? top : left, *side = vert ? left : top;
    const int pos = (v + 1) * a, i = pos >> 5, f = pos & 31;
    int idx = u + i + 1;                               /* main[idx]; main[0] = top-left, main[k] = mainr[k-1] */
    int s0, s1;
    if (idx > 0)
        s0 = mainr[idx - 1];
    else if (idx == 0)
        s0 = tl;
    else
        s0 = side[((-idx * c_inv[-d] + 128) >> 8) - 1];
    idx++;
    if (idx > 0)
        s1 = mainr[idx - 1];
    else if (idx == 0)
        s1 = tl;
    else
        s1 = side[((-idx * c_inv[-d] + 128) >> 8) - 1];
    return ((32 - f) * s0 + f * s1 + 16) >> 5;
}

/* wave sum on the DPP path (quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror: after the quad steps a mirror step completes the next power of two) and
 * v_readlane across the four rows: __shfl_xor compiles to ds_bpermute_b32, an LDS-crossbar round trip per step, and every (coding unit, mode) task of this kernel - a whole
 * wave for as few as 64 samples - ends in one.  Wave-uniform result. */
#define OIS_DPP(v, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    v += OIS_DPP(v, 0xB1);
    v += OIS_DPP(v, 0x4E);
    v += OIS_DPP(v, 0x141);
    v += OIS_DPP(v, 0x140);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) + (uint32_t)__builtin_amdgcn_readlane((int)v, 32) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

/* SAD of CU `cu` predicted with `mode`; executed by one whole wavefront */
__device__ __forceinline__ uint32_t task_sad(const OisShared &S, int cu, int mode, int lane)
{
    int cx, cy, N, lg;
    cu_geom(cu, cx, cy, N, lg);
    const uint8_t *r = S.refs + ref_base(cu);
    const int dc = S.dc[cu];
    uint32_t acc = 0;
    for (int p = lane; p < N * N; p += 64) {
        const int y = p >> lg, x = p & (N - 1);
        const int pr = predict_sample(mode, N, lg, r, x, y, dc);
        acc += (uint32_t)abs((int)S.win[(cy + y + 1) * WIN_W + cx + x + WIN_X0] - pr);
    }
    return wave_sum(acc);
}

/* DC prediction is a constant except for the filtered first row / column (N < 32): SAD four samples per lane-op
 * against the replicated DC value; the DC SAD of every CU is the one task every P/B picture always runs */
__device__ __forceinline__ uint32_t task_sad_dc(const OisShared &S, int cu, int lane)
{
    int cx, cy, N, lg;
    cu_geom(cu, cx, cy, N, lg);
    const uint8_t *r = S.refs + ref_base(cu);
    const int dc = S.dc[cu];
    const uint32_t dc4 = (uint32_t)dc * 0x01010101u;
    const int qshift = lg - 2; /* dwords per row = N / 4 */
    uint32_t acc = 0;
    for (int q = lane; q < (N * N) >> 2; q += 64) {
        const int y = q >> qshift, x4 = (q & ((1 << qshift) - 1)) << 2;
        const uint32_t sv = *(const uint32_t *)&S.win[(cy + y + 1) * WIN_W + cx + x4 + WIN_X0];
        uint32_t pv = dc4;
        if (N < 32 && (y == 0 || x4 == 0)) {
            pv = 0;
#pragma unroll
            for (int i = 0; i < 4; i++)
                pv |= (uint32_t)predict_sample(1, N, lg, r, x4 + i, y, dc) << (8 * i);
        }
        acc = __builtin_amdgcn_sad_u8(sv, pv, acc);
    }
    return wave_sum(acc);
}

#define W_DIST (1u << 21)
#define W_VALID (1u << 22)
#define W_MODE (1u << 23)
__device__ __forceinline__ void set_dist(uint32_t &c, uint32_t d) { c = (c & ~0xFFFFFu) | (d & 0xFFFFFu) | W_DIST; }
__device__ __forceinline__ void set_valid(uint32_t &c, int v) { c = (c & ~(1u << 20)) | ((uint32_t)(v != 0) << 20) | W_VALID; }
__device__ __forceinline__ void set_mode(uint32_t &c, uint32_t m) { c = (c & 0x00FFFFFFu) | ((m & 0xFFu) << 24) | W_MODE; }

/* Decision step of one CU given its SADs (S.sad[cu][k], k = position in the tested mode list).
 * bestMode / stage1 carry the reference's function-scope state; returns false when the search of this CU did not
 * update bestMode (the caller must then replay serially). */
__device__ bool decide_cu(OisShared &S, const SvtAmdOisParams &P, int cu, bool valid, uint32_t meSad, uint32_t &bestMode,
                          uint32_t *stage1)
{
    uint32_t *cand = S.out_cand[cu];
    for (int k = 0; k < SVT_AMD_OIS_MAX_CAND; k++)
        cand[k] = 0;
    S.out_total[cu] = 0xFF;
    int cx, cy, N, lg;
    cu_geom(cu, cx, cy, N, lg);
    bool updated = true;
    if (P.slice_is_intra) {
        for (int k = 0; k < 7; k++)
            set_valid(cand[k], 0);
        if (!valid)
            return true;
        if (N == 32) {
            set_dist(cand[0], S.sad[cu][0]);
            set_mode(cand[0], 0);
            set_valid(cand[0], 1);
            return true;
        }
        uint32_t best = 32 * 32 * 255;
        updated = false;
        for (int k = 0; k < 7; k++) {
            stage1[k] = S.sad[cu][k];
            if (stage1[k] < best)
                bestMode = c_islice[k], best = stage1[k], updated = true;
        }
        int count = 0;
        set_valid(cand[0], 1);
        set_dist(cand[0], stage1[0]);
        set_mode(cand[count++], 0);
        set_mode(cand[count++], 1);
        if (bestMode > 1) {
            const int g = bestMode == 2 ? 0 : bestMode == 10 ? 1 : bestMode == 18 ? 2 : bestMode == 26 ? 3 : 4;
            for (int k = 0; k < 3; k++)
                set_mode(cand[count++], c_isl_inject[g][k]);
        }
        S.out_total[cu] = (uint8_t)count;
        return updated;
    }
    if (!valid)
        return true;
    if (P.ois_kernel_level) {
        for (int k = 0; k < 18; k++)
            set_valid(cand[k], 0);
        for (uint32_t m = 0; m < 35; m++) {
            const uint32_t sad = S.sad[cu][m];
            if (m < 18) {
                set_dist(cand[m], sad);
                set_mode(cand[m], m);
            } else {
                uint32_t worst = cand[0] & 0xFFFFFu, wi = 0;
                for (uint32_t k = 1; k < 18; k++)
                    if ((cand[k] & 0xFFFFFu) > worst)
                        worst = cand[k] & 0xFFFFFu, wi = k;
                if (sad < worst) {
                    set_dist(cand[wi], sad);
                    set_mode(cand[wi], m);
                }
            }
        }
        for (int i = 0; i < 18; i++)
            for (int j = i; j < 18; j++)
                if ((cand[i] & 0xFFFFFu) > (cand[j] & 0xFFFFFu)) {
                    const uint32_t mi = cand[i] >> 24, di = cand[i] & 0xFFFFFu;
                    set_mode(cand[i], cand[j] >> 24);
                    set_mode(cand[j], mi);
                    set_dist(cand[i], cand[j] & 0xFFFFFu);
                    set_dist(cand[j], di);
                }
        S.out_total[cu] = 18;
        return true;
    }
    for (int k = 0; k < 9; k++)
        set_valid(cand[k], 0);
    if (P.limit_ois_to_dc_mode) {
        set_dist(cand[0], S.sad[cu][9]);
        set_mode(cand[0], 1);
        set_valid(cand[0], 1);
        S.out_total[cu] = 1;
        return true;
    }
    stage1[0] = S.sad[cu][9]; /* DC SAD (slot 9) */
    const int n = S.nmodes[cu];
    if (n == 0) {
        set_mode(cand[0], 1);
        set_dist(cand[0], stage1[0]);
        S.out_total[cu] = 1;
        return true;
    }
    (void)meSad;
    uint32_t best = 32 * 32 * 255;
    updated = false;
    for (int k = 0; k < n; k++) {
        stage1[k] = S.sad[cu][k];
        if (stage1[k] < best)
            bestMode = c_stage1[k], best = stage1[k], updated = true;
    }
    int g = 8;
    for (int k = 0; k < 8; k++)
        if (bestMode == c_stage1[k])
            g = k;
    set_dist(cand[0], stage1[g]);
    set_valid(cand[0], P.set_best_ois_distortion_to_valid);
    for (int k = 0; k < 9; k++)
        set_mode(cand[k], c_inject[g][k]);
    S.out_total[cu] = (uint8_t)n;
    return updated;
}

__global__ __launch_bounds__(256) void k_ois_picture(const OisJobDev *__restrict__ jobs)
{
    __shared__ OisShared S;
    const OisJobDev &J = jobs[blockIdx.y];
    const int lcu = blockIdx.x;
    if (lcu >= J.nlcu)
        return;
    const SvtAmdOisParams P = J.P;
    const uint8_t *__restrict__ full = J.full;
    const int pitch = J.pitch, lcus_w = J.lcus_w;
    const SvtAmdMeLcuResult *__restrict__ me = J.me;
    SvtAmdOisLcuResult *__restrict__ out = J.out;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lx = (lcu % lcus_w) * 64, ly = (lcu / lcus_w) * 64;
    const int W = P.luma_width, H = P.luma_height;
    const int last = P.slice_is_intra ? 84 : ((P.skip_ois_8x8 || P.cu8x8_mode == 1) ? 20 : 84);

    /* 1. window (the padded plane makes every address valid; out-of-picture samples are never USED); rows start at
     * LCU column -4, a dword boundary of the plane, so every load is one aligned dword */
    for (int i = t; i < WIN_H * 25; i += 256) {
        const int row = i / 25, c4 = i - row * 25;
        *(uint32_t *)&S.win[row * WIN_W + c4 * 4] =
            *(const uint32_t *)(full + (ptrdiff_t)(ly + row - 1) * pitch + lx - WIN_X0 + c4 * 4);
    }
    if (t == 0)
        S.stale = 0;
    __syncthreads();

    /* 2. reference arrays of every CU */
    const int nref = last == 20 ? 516 + 1040 : 516 + 1040 + 2112; /* no 8x8 CUs when last == 20 */
    for (int i = t; i < nref; i += 256) {
        int cu, e;
        if (i < 516)
            cu = 1 + i / 129, e = i % 129;
        else if (i < 1556)
            cu = 5 + (i - 516) / 65, e = (i - 516) % 65;
        else
            cu = 21 + (i - 1556) / 33, e = (i - 1556) % 33;
        int cx, cy, N, lg;
        cu_geom(cu, cx, cy, N, lg);
        const int ox = lx + cx, oy = ly + cy;
        int v = 128;
        if (e < 2 * N) {
            if (ox != 0 && oy + e < H)
                v = S.win[(cy + e + 1) * WIN_W + cx + WIN_X0 - 1];
        } else if (e == 2 * N) {
            if (ox != 0 && oy != 0)
                v = S.win[cy * WIN_W + cx + WIN_X0 - 1];
        } else {
            const int j = e - 2 * N - 1;
            if (oy != 0 && ox + j < W)
                v = S.win[cy * WIN_W + cx + j + WIN_X0];
        }
        S.refs[i] = (uint8_t)v;
    }
    __syncthreads();
    if (t >= 1 && t <= last) {
        int cx, cy, N, lg;
        cu_geom(t, cx, cy, N, lg);
        const uint8_t *r = S.refs + ref_base(t);
        uint32_t s = 0;
        for (int k = 0; k < N; k++)
            s += r[k] + r[2 * N + 1 + k];
        S.dc[t] = (uint8_t)((s + N) >> (lg + 1));
        S.nmodes[t] = 0;
    }
    __syncthreads();

#define CU_VALID(cu_, v_)                                                          \
    do {                                                                           \
        int cx_, cy_, N_, lg_;                                                     \
        cu_geom(cu_, cx_, cy_, N_, lg_);                                           \
        v_ = !(lx + cx_ + N_ > W || ly + cy_ + N_ > H);                            \
    } while (0)

    /* 3. SAD tasks */
    if (P.slice_is_intra) {
        for (int task = wave; task < 4 + 80 * 7; task += 4) {
            const int cu = task < 4 ? 1 + task : 5 + (task - 4) / 7, k = task < 4 ? 0 : (task - 4) % 7;
            bool valid;
            CU_VALID(cu, valid);
            if (!valid)
                continue;
            const uint32_t s = task_sad(S, cu, c_islice[k], lane);
            if (lane == 0)
                S.sad[cu][k] = s;
        }
    } else if (P.ois_kernel_level) {
        for (int task = wave; task < last * 35; task += 4) {
            const int cu = 1 + task / 35, m = task % 35;
            bool valid;
            CU_VALID(cu, valid);
            if (!valid)
                continue;
            const uint32_t s = task_sad(S, cu, m, lane);
            if (lane == 0)
                S.sad[cu][m] = s;
        }
    } else {
        for (int cu = 1 + wave; cu <= last; cu += 4) { /* DC of every CU -> slot 9 */
            bool valid;
            CU_VALID(cu, valid);
            if (!valid)
                continue;
            const uint32_t s = task_sad_dc(S, cu, lane);
            if (lane == 0)
                S.sad[cu][9] = s;
        }
        __syncthreads();
        if (!P.limit_ois_to_dc_mode) {
            if (t >= 1 && t <= last) { /* GetInterIntraSadDistance / GetOisPoint (EbMotionEstimation.c:4782,4814) */
                bool valid;
                CU_VALID(t, valid);
                if (valid) {
                    const uint32_t meSad = me[lcu].pu[t].distortion[0], dcSad = S.sad[t][9];
                    const int32_t diff = (int32_t)((meSad - dcSad) * 100u);
                    const int32_t dist = dcSad ? diff / (int32_t)dcSad : 0;
                    int point = 4;
                    const int16_t *th = c_ois_th[P.ois_th_set][P.temporal_layer_index];
                    if (dcSad == 0 || meSad == 0 || dist <= th[0])
                        point = 0;
                    else if (dist <= th[1])
                        point = 1;
                    else if (dist <= th[2])
                        point = 2;
                    else if (dist <= th[3])
                        point = 3;
                    S.nmodes[t] = (uint8_t)(point == 0 ? 0 : 2 * point + 1);
                }
            }
            __syncthreads();
            for (int task = wave; task < last * 9; task += 4) {
                const int cu = 1 + task / 9, k = task % 9;
                if (k >= S.nmodes[cu])
                    continue;
                const uint32_t s = task_sad(S, cu, c_stage1[k], lane);
                if (lane == 0)
                    S.sad[cu][k] = s;
            }
        }
    }
    __syncthreads();

    /* 4. decisions: one thread per CU, then the serial replay if the carried state mattered */
    if (t >= 1 && t <= 84) {
        if (t <= last) {
            bool valid;
            CU_VALID(t, valid);
            uint32_t bm = 0, st[11];
            for (int k = 0; k < 11; k++)
                st[k] = 0;
            if (!decide_cu(S, P, t, valid, 0, bm, st))
                S.stale = 1;
        } else {
            for (int k = 0; k < SVT_AMD_OIS_MAX_CAND; k++)
                S.out_cand[t][k] = 0;
            S.out_total[t] = 0xFF;
        }
    }
    if (t == 0) {
        for (int k = 0; k < SVT_AMD_OIS_MAX_CAND; k++)
            S.out_cand[0][k] = 0;
        S.out_total[0] = 0xFF;
    }
    __syncthreads();
    if (S.stale && t == 0) {
        uint32_t bm = 0, st[11];
        for (int k = 0; k < 11; k++)
            st[k] = 0;
        for (int cu = 1; cu <= last; cu++) {
            bool valid;
            CU_VALID(cu, valid);
            decide_cu(S, P, cu, valid, 0, bm, st);
        }
    }
    __syncthreads();

    /* 5. write the record */
    uint32_t *o = (uint32_t *)&out[lcu];
    const uint32_t *c = &S.out_cand[0][0];
    for (int i = t; i < 85 * SVT_AMD_OIS_MAX_CAND; i += 256)
        o[i] = c[i];
    uint8_t *ot = out[lcu].total_intra_luma_mode;
    if (t < 88)
        ot[t] = t < 85 ? S.out_total[t] : 0;
}

int svt_amd_launch_ois_batch(SvtAmdContext *ctx, const OisJobDev *host_jobs, int njobs, int max_lcus)
{
    {
        const int rcd = svt_amd_upload_descriptors(ctx, ctx->d_ois_jobs, host_jobs, sizeof(OisJobDev) * (size_t)njobs);
        if (rcd)
            return rcd;
    }
    hipLaunchKernelGGL(k_ois_picture, dim3(max_lcus, njobs), dim3(256), 0, ctx->stream, (const OisJobDev *)ctx->d_ois_jobs);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* ---------------------------------------------------------------------------------------------------
 * ComputeDecimatedZzSad (EbMotionEstimationProcess.c:176-300): one wavefront per LCU; the previous picture's
 * 1/16 plane IS the collocated LCU decimated by 4 (Decimation2D is a point sub-sampler), so both operands
 * come from the planes prep already built.  Lane = (row, 4-sample group): one v_sad_u8.
 * --------------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void k_zz_sad(const uint8_t *__restrict__ cur16, const uint8_t *__restrict__ prev16,
                                                int pitch, int width, int height, int nlcu, int lcus_w,
                                                SvtAmdZzLcu *__restrict__ out)
{
    const int lcu = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (lcu >= nlcu)
        return;
    const int ox = (lcu % lcus_w) * 64, oy = (lcu / lcus_w) * 64;
    const int lw = min(64, width - ox), lh = min(64, height - oy);
    uint32_t sad = ~0u;
    uint8_t zz = 0xFF;
    if (lw == 64 && lh == 64) {
        const int r = lane >> 2, g = (lane & 3) << 2;
        const ptrdiff_t at = (ptrdiff_t)((oy >> 2) + r) * pitch + (ox >> 2) + g;
        const uint32_t s = __builtin_amdgcn_sad_u8(*(const uint32_t *)(cur16 + at), *(const uint32_t *)(prev16 + at), 0u);
        sad = wave_sum(s);
        zz = sad < 256 ? 0 : sad < 512 ? 3 : sad < 1024 ? 10 : sad < 2048 ? 20 : 30;
    }
    if (lane == 0) {
        const uint32_t area = (uint32_t)((lw >> 2) * (lh >> 2));
        SvtAmdZzLcu o;
        o.sad = sad, o.zz_cost = zz;
        o.non_moving_index = sad < area * 2 ? 0 : sad < area * 4 ? 10 : sad < area * 8 ? 20 : 30;
        o.pad[0] = o.pad[1] = 0;
        out[lcu] = o;
    }
}

int svt_amd_launch_zz_sad(SvtAmdContext *ctx, const DevPicture *cur, const DevPicture *prev, SvtAmdZzLcu *d_out)
{
    const int lw = (cur->width + 63) / 64, lh = (cur->height + 63) / 64;
    hipLaunchKernelGGL(k_zz_sad, dim3((lw * lh + 3) / 4), dim3(256), 0, ctx->stream, (const uint8_t *)cur->sixteenth.origin,
                       (const uint8_t *)prev->sixteenth.origin, (int)cur->sixteenth.pitch, (int)cur->width, (int)cur->height,
                       lw * lh, lw, d_out);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
