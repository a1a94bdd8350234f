/*
 * oracle/svt_oracle_ois.c - TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 * CPU restatement of the reference's open-loop intra search for one LCU:
 *   OpenLoopIntraSearchLcu                      Codec/EbMotionEstimation.c:5053-5320
 *   IntraOpenLoopSearchTheseModesOutputBest     :4458-4498
 *   InjectIntraCandidatesBasedOnBestModeIslice  :4500-4560, ...BasedOnBestMode :4562-4780
 *   GetInterIntraSadDistance :4782, GetOisPoint :4814, SortIntraModesOpenLoop :4873,
 *   SortOisCandidateOpenLoop :4843, OpenLoopIntraDC :4965
 *   UpdateNeighborSamplesArrayOpenLoop          Codec/EbIntraPrediction.c:5222-5334
 *   IntraPredictionOpenLoop :5339-5421, IntraModeAngular_all :3516-3594 and its four group helpers
 * Pinned against the reference by tests/test_oracle_ois_golden.py (before/after dumps of real encoder runs).
 *
 * Output convention (include/svt_hevc_amd.h, SvtAmdOisLcuResult): starts from zero / 0xFF and marks every
 * bitfield the reference wrote with SVT_AMD_OIS_W_*; the reference leaves the other bits as the pool held them.
 */
#include <string.h>
#include "svt_oracle.h"

#define W_DIST (1u << 21)
#define W_VALID (1u << 22)
#define W_MODE (1u << 23)

static inline void set_dist(uint32_t *c, uint32_t d) { *c = (*c & ~0xFFFFFu) | (d & 0xFFFFFu) | W_DIST; }
static inline void set_valid(uint32_t *c, int v) { *c = (*c & ~(1u << 20)) | ((uint32_t)(v != 0) << 20) | W_VALID; }
static inline void set_mode(uint32_t *c, uint32_t m) { *c = (*c & 0x00FFFFFFu) | ((m & 0xFFu) << 24) | W_MODE; }
static inline uint32_t get_dist(uint32_t c) { return c & 0xFFFFFu; }
static inline uint32_t get_mode(uint32_t c) { return c >> 24; }

/* RASTER_SCAN_CU_X/Y/SIZE/DEPTH (Codec/EbUtility.c) for index 1..84 */
static void cu_geom(int idx, int *x, int *y, int *size, int *depth)
{
    if (idx < 5)
        *size = 32, *depth = 1, *x = ((idx - 1) & 1) * 32, *y = ((idx - 1) >> 1) * 32;
    else if (idx < 21)
        *size = 16, *depth = 2, *x = ((idx - 5) & 3) * 16, *y = ((idx - 5) >> 2) * 16;
    else
        *size = 8, *depth = 3, *x = ((idx - 21) & 7) * 8, *y = ((idx - 21) >> 3) * 8;
}

typedef struct OisCtx {
    uint8_t rev[4 * 32 + 1]; /* yIntraReferenceArrayReverse: left top-to-bottom, top-left, top */
    uint8_t fwd[4 * 32 + 1]; /* yIntraReferenceArray: left bottom-to-top, top-left, top        */
    uint8_t pred[32 * 64];   /* meContextPtr->lcuBuffer, stride MAX_LCU_SIZE                    */
} OisCtx;

/* EbIntraPrediction.c:5222 - source neighbours, 128 where outside the picture; no substitution, no smoothing */
static void update_neighbors(OisCtx *c, const uint8_t *luma, uint32_t stride, uint32_t width, uint32_t height,
                             uint32_t ox, uint32_t oy, uint32_t N)
{
    const uint32_t N2 = N << 1;
    const uint8_t *src = luma + (size_t)oy * stride + ox;
    memset(c->rev, 128, 4 * N + 1);
    if (ox != 0) {
        const uint32_t cnt = (oy + N2 > height) ? N2 - (oy + N2 - height) : N2;
        for (uint32_t i = 0; i < cnt; i++)
            c->rev[i] = src[(ptrdiff_t)i * stride - 1];
    }
    if (ox != 0 && oy != 0)
        c->rev[N2] = src[-(ptrdiff_t)stride - 1];
    if (oy != 0) {
        const uint32_t cnt = (ox + N2 > width) ? N2 - (ox + N2 - width) : N2;
        memcpy(c->rev + N2 + 1, src - stride, cnt);
    }
    memcpy(c->fwd + N2, c->rev + N2, N2 + 1);
    for (uint32_t i = 0; i < N2; i++)
        c->fwd[N2 - 1 - i] = c->rev[i];
}

static const int32_t ANG[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};           /* intraModeAngularTable :25 */
static const uint32_t INV[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256}; /* invIntraModeAngularTable :48 */

/* IntraPredictionOpenLoop :5339 -> prediction of `mode` into c->pred (stride 64) */
static void predict(OisCtx *c, uint32_t N, uint32_t mode)
{
    uint8_t line[3 * 32 + 8];
    if (mode == 0)
        svt_oracle_IntraPred(SVT_ORACLE_INTRA_PLANAR, 1, N, c->rev, c->pred, 64, 0, 0);
    else if (mode == 1)
        svt_oracle_IntraPred(SVT_ORACLE_INTRA_DC_LUMA, 1, N, c->rev, c->pred, 64, 0, 0);
    else if (mode == 26)
        svt_oracle_IntraPred(SVT_ORACLE_INTRA_VERTICAL_LUMA, 1, N, c->rev, c->pred, 64, 0, 0);
    else if (mode == 10)
        svt_oracle_IntraPred(SVT_ORACLE_INTRA_HORIZONTAL_LUMA, 1, N, c->rev, c->pred, 64, 0, 0);
    else if (mode == 34)
        svt_oracle_IntraPred(SVT_ORACLE_INTRA_ANGULAR_34, 1, N, c->fwd, c->pred, 64, 0, 0);
    else if (mode == 18)
        svt_oracle_IntraPred(SVT_ORACLE_INTRA_ANGULAR_18, 1, N, c->fwd, c->pred, 64, 0, 0);
    else if (mode == 2)
        svt_oracle_IntraPred(SVT_ORACLE_INTRA_ANGULAR_2, 1, N, c->rev, c->pred, 64, 0, 0);
    else if (mode >= 27) /* IntraModeAngular_27To33 :3180 */
        svt_oracle_IntraPred(SVT_ORACLE_INTRA_ANGULAR_VERTICAL, 1, N, c->fwd + 2 * N, c->pred, 64, 0, ANG[mode - 26]);
    else if (mode >= 19) { /* IntraModeAngular_19To25 :3222 */
        const int32_t angle = -ANG[26 - mode];
        uint32_t invSum = 128;
        uint8_t *mainp = line + 32 + (N - 1) - (N - 1); /* refAbove + (size-1), with room for negative indices */
        mainp = line + 40;
        for (uint32_t i = 0; i < N + 1; i++)
            mainp[i] = c->fwd[2 * N + i];
        for (int32_t s = -1; s > ((int32_t)N * angle >> 5); --s) {
            invSum += INV[26 - mode];
            mainp[s] = c->fwd[2 * N - (int32_t)(invSum >> 8)];
        }
        svt_oracle_IntraPred(SVT_ORACLE_INTRA_ANGULAR_VERTICAL, 1, N, mainp, c->pred, 64, 0, angle);
    } else if (mode >= 11) { /* IntraModeAngular_11To17 :3349 */
        const int32_t angle = -ANG[mode - 10];
        uint32_t invSum = 128;
        uint8_t *mainp = line + 40;
        for (uint32_t i = 0; i < N + 1; i++)
            mainp[i] = c->fwd[2 * N - i];
        for (int32_t s = -1; s > ((int32_t)N * angle >> 5); --s) {
            invSum += INV[mode - 10];
            mainp[s] = c->fwd[2 * N + (invSum >> 8)];
        }
        svt_oracle_IntraPred(SVT_ORACLE_INTRA_ANGULAR_HORIZONTAL, 1, N, mainp, c->pred, 64, 0, angle);
    } else /* 3..9: IntraModeAngular_3To9 :3454 */
        svt_oracle_IntraPred(SVT_ORACLE_INTRA_ANGULAR_HORIZONTAL, 1, N, c->rev - 1, c->pred, 64, 0, ANG[10 - mode]);
}

static uint32_t sad_vs_pred(const OisCtx *c, const uint8_t *src, uint32_t stride, uint32_t N)
{
    uint32_t s = 0;
    for (uint32_t y = 0; y < N; y++)
        for (uint32_t x = 0; x < N; x++) {
            const int d = (int)src[(size_t)y * stride + x] - (int)c->pred[y * 64 + x];
            s += (uint32_t)(d < 0 ? -d : d);
        }
    return s;
}

static const uint32_t ISLICE_MODES[7] = {0, 1, 10, 26, 2, 18, 34};            /* iSliceModesArray :148 (first MAX_OIS_0) */
static const uint32_t STAGE1_MODES[9] = {10, 26, 2, 18, 34, 6, 14, 22, 30};   /* stage1ModesArray :149 */
static const int32_t OIS_TH[3][6][4] = {                                        /* EbHevcOisPointTh :26 */
    {{-20, 50, 150, 200}, {-20, 50, 150, 200}, {-20, 50, 100, 150}, {-20, 50, 200, 300}, {-20, 50, 200, 300}, {-20, 50, 200, 300}},
    {{-150, 0, 150, 200}, {-150, 0, 150, 200}, {-125, 0, 100, 150}, {-50, 50, 200, 300}, {-50, 50, 200, 300}, {-50, 50, 200, 300}},
    {{-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}}};
static const uint8_t N_MODES_AT_POINT[5] = {1, 3, 5, 7, 9}; /* numberOfOisModePoints :85; totalIntraLumaMode[p][depth>=1] is the same */

/* the nine 9-entry candidate lists of InjectIntraCandidatesBasedOnBestMode, by stage1 index */
static const uint8_t INJECT[9][9] = {
    {10, 1, 0, 9, 11, 8, 12, 7, 13},    /* H   (stage1SadArray[0]) */
    {26, 1, 0, 25, 27, 24, 28, 23, 29}, /* V   [1] */
    {2, 1, 0, 3, 4, 5, 7, 8, 9},        /* 2   [2] */
    {18, 1, 0, 17, 19, 16, 20, 15, 21}, /* 18  [3] */
    {34, 1, 0, 33, 32, 29, 31, 27, 28}, /* 34  [4] */
    {6, 1, 0, 7, 5, 4, 8, 3, 9},        /* 6   [5] */
    {14, 1, 0, 13, 15, 12, 16, 11, 17}, /* 14  [6] */
    {22, 1, 0, 21, 23, 20, 24, 19, 25}, /* 22  [7] */
    {30, 1, 0, 29, 31, 28, 32, 27, 33}, /* 30 and default [8] */
};

void svt_oracle_ois_lcu(const SvtAmdOisParams *P, const uint8_t *luma, uint32_t stride, uint32_t lcu_x, uint32_t lcu_y,
                        const uint32_t *me_sad /* [85] or NULL */, SvtAmdOisLcuResult *out)
{
    OisCtx c;
    uint32_t stage1[11] = {0};
    uint32_t bestMode = 0; /* EB_INTRA_PLANAR; persists across the CUs of the LCU */
    memset(out, 0, sizeof(*out));
    memset(out->total_intra_luma_mode, 0xFF, sizeof(out->total_intra_luma_mode));
    const uint32_t W = P->luma_width, H = P->luma_height;
    const int last = P->slice_is_intra ? 84 : ((P->skip_ois_8x8 || P->cu8x8_mode == 1) ? 20 : 84);

    for (int cu = 1; cu <= last; cu++) {
        int cx, cy, size, depth;
        cu_geom(cu, &cx, &cy, &size, &depth);
        const uint32_t N = (uint32_t)size, ox = lcu_x + cx, oy = lcu_y + cy;
        const int valid = !(ox + N > W || oy + N > H); /* rasterScanCuValidity, EbSequenceControlSet.c:277 */
        uint32_t *cand = out->candidate[cu];
        const uint8_t *src = luma + (size_t)oy * stride + ox;

        if (P->slice_is_intra) {
            for (int k = 0; k < 7; k++)
                set_valid(&cand[k], 0);
            if (!valid)
                continue;
            update_neighbors(&c, luma, stride, W, H, ox, oy, N);
            if (N == 32) {
                predict(&c, N, 0);
                set_dist(&cand[0], sad_vs_pred(&c, src, stride, N));
                set_mode(&cand[0], 0);
                set_valid(&cand[0], 1);
                continue; /* totalIntraLumaMode untouched (:5100-5122) */
            }
            uint32_t best = 32 * 32 * 255;
            for (int k = 0; k < 7; k++) {
                predict(&c, N, ISLICE_MODES[k]);
                stage1[k] = sad_vs_pred(&c, src, stride, N);
                if (stage1[k] < best)
                    bestMode = ISLICE_MODES[k], best = stage1[k];
            }
            /* InjectIntraCandidatesBasedOnBestModeIslice :4500 */
            int count = 0;
            set_valid(&cand[0], 1);
            set_dist(&cand[0], stage1[0]);
            set_mode(&cand[count++], 0);
            set_mode(&cand[count++], 1);
            static const uint8_t ISL[5][3] = {{2, 4, 6}, {10, 6, 14}, {18, 14, 22}, {26, 22, 30}, {34, 32, 30}};
            if (bestMode > 1) {
                const int g = bestMode == 2 ? 0 : bestMode == 10 ? 1 : bestMode == 18 ? 2 : bestMode == 26 ? 3 : 4;
                for (int k = 0; k < 3; k++)
                    set_mode(&cand[count++], ISL[g][k]);
            }
            out->total_intra_luma_mode[cu] = (uint8_t)count;
            continue;
        }

        if (!valid)
            continue;
        if (!P->limit_ois_to_dc_mode)
            update_neighbors(&c, luma, stride, W, H, ox, oy, N);

        if (P->ois_kernel_level) {
            for (int k = 0; k < 18; k++)
                set_valid(&cand[k], 0);
            for (uint32_t m = 0; m < 35; m++) {
                predict(&c, N, m);
                const uint32_t sad = sad_vs_pred(&c, src, stride, N);
                if (m < 18) { /* SortIntraModesOpenLoop :4873 */
                    set_dist(&cand[m], sad);
                    set_mode(&cand[m], m);
                } else {
                    uint32_t worst = get_dist(cand[0]), wi = 0;
                    for (uint32_t k = 1; k < 18; k++)
                        if (get_dist(cand[k]) > worst)
                            worst = get_dist(cand[k]), wi = k;
                    if (sad < worst) {
                        set_dist(&cand[wi], sad);
                        set_mode(&cand[wi], m);
                    }
                }
            }
            for (int i = 0; i < 18; i++) /* SortOisCandidateOpenLoop :4843 */
                for (int j = i; j < 18; j++)
                    if (get_dist(cand[i]) > get_dist(cand[j])) {
                        const uint32_t mi = get_mode(cand[i]), di = get_dist(cand[i]);
                        set_mode(&cand[i], get_mode(cand[j]));
                        set_mode(&cand[j], mi);
                        set_dist(&cand[i], get_dist(cand[j]));
                        set_dist(&cand[j], di);
                    }
            out->total_intra_luma_mode[cu] = 18;
            continue;
        }

        for (int k = 0; k < 9; k++)
            set_valid(&cand[k], 0);
        if (P->limit_ois_to_dc_mode) { /* OpenLoopIntraDC :4965 */
            update_neighbors(&c, luma, stride, W, H, ox, oy, N);
            predict(&c, N, 1);
            set_dist(&cand[0], sad_vs_pred(&c, src, stride, N));
            set_mode(&cand[0], 1);
            set_valid(&cand[0], 1);
            out->total_intra_luma_mode[cu] = 1;
            continue;
        }
        const uint32_t meSad = me_sad ? me_sad[cu] : 0;
        predict(&c, N, 1); /* GetInterIntraSadDistance :4782 */
        stage1[0] = sad_vs_pred(&c, src, stride, N);
        const int32_t sadDiff = (int32_t)(meSad - stage1[0]) * 100;
        const int32_t dist = stage1[0] ? sadDiff / (int32_t)stage1[0] : 0;
        int point = 4; /* GetOisPoint :4814 */
        const int32_t *th = OIS_TH[P->ois_th_set][P->temporal_layer_index];
        if (stage1[0] == 0 || meSad == 0 || dist <= th[0])
            point = 0;
        else if (dist <= th[1])
            point = 1;
        else if (dist <= th[2])
            point = 2;
        else if (dist <= th[3])
            point = 3;
        if (point == 0) {
            set_mode(&cand[0], 1);
            set_dist(&cand[0], stage1[0]);
            out->total_intra_luma_mode[cu] = 1;
            continue;
        }
        const int n = N_MODES_AT_POINT[point];
        uint32_t best = 32 * 32 * 255;
        for (int k = 0; k < n; k++) {
            predict(&c, N, STAGE1_MODES[k]);
            stage1[k] = sad_vs_pred(&c, src, stride, N);
            if (stage1[k] < best)
                bestMode = STAGE1_MODES[k], best = stage1[k];
        }
        int g = 8; /* InjectIntraCandidatesBasedOnBestMode :4562 (MODE_30 and default share a list) */
        for (int k = 0; k < 8; k++)
            if (bestMode == STAGE1_MODES[k])
                g = k;
        set_dist(&cand[0], stage1[g]);
        set_valid(&cand[0], P->set_best_ois_distortion_to_valid);
        for (int k = 0; k < 9; k++)
            set_mode(&cand[k], INJECT[g][k]);
        out->total_intra_luma_mode[cu] = N_MODES_AT_POINT[point];
    }
}
