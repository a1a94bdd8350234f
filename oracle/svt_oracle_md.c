/*
 * oracle/svt_oracle_md.c - TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 * CPU checker of the device-resident mode decision (include/svt_hevc_amd.h "Device-resident mode decision"): ModeDecisionLcu
 * (Codec/EbProductCodingLoop.c:4691-5114) for the pictures that revision covers (I pictures, PICT_FULL84, closed-loop intra, luma-only
 * candidates), as a composite of
 *   - svt-hevc_amd/csrc/md_logic.h: the scalar decisions (context generation, candidate lists, fast cost, candidate buffers, pre-mode
 *     decision, full cost, full mode decision, partition exit, inter-depth decision) - ONE text shared with the HIP kernel, so that the
 *     control code the GPU executes is the code this file runs on the CPU against the reference's own records;
 *   - the pinned oracle leaves for everything that touches samples: svt_oracle_intra_pu (GenerateIntraLumaReferenceSamplesMd +
 *     IntraPredictionCl, tests/test_oracle_intramd_golden.py), svt_oracle_NxMSadKernel (tests/test_oracle_leaf.py),
 *     svt_oracle_product_full_loop_luma (ProductFullLoop, tests/test_oracle_fullloop_golden.py), svt_oracle_recon_tu
 *     (PerformInverseTransformRecon: EstimateInvTransform + addition; the C tables hold the same inverse transforms as the encode
 *     pass, Codec/EbTransforms.h:474-510).
 * The mode decision's neighbour arrays (pcs->md*NeighborArray[MD_NEIGHBOR_ARRAY_INDEX]: candidate reconstruction, mode type, intra luma
 * mode, depth, skip flag) are kept as picture-sized maps written with the same values in the same order
 * (ModeDecisionUpdateNeighborArrays, :371): an array entry's last writer is the unit that holds the neighbouring position, as for the
 * encode pass (svt_oracle_encodepass.c).
 * PINNED by tests/test_oracle_md_golden.py on recorded ModeDecisionLcu calls of whole pictures (tests/golden/md_*.npz: split flags,
 * modes, luma cbf and the costs of every tested leaf, oracle/ref_harness_md_dump.c).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "svt_oracle.h"
#include "../svt-hevc_amd/csrc/md_logic.h"

typedef struct MdPic {
    uint8_t *rec;      /* the mode decision's luma reconstruction of the units decided so far (mdLumaReconNeighborArray) */
    uint32_t pitch;
    uint32_t *info;    /* per 4x4 luma block: mode type | intra luma mode << 8 | depth << 16 | skip flag << 24; ~0 = never written */
    uint32_t infoPitch;
    uint32_t w, h;
    MdMvUnit *mv;      /* per 8x8 luma block: the unit's MvUnit_t (mdMvNeighborArray) */
    uint32_t mvPitch;
    /* P / B pictures */
    const SvtAmdMdInter *X;
    const SvtAmdMeLcuResult *me;
    const SvtAmdTmvpLcu *tmvp;
    const SvtAmdRefPicture *ref[2];
    const uint8_t *src, *src_c[2];
    uint32_t srcStride, srcStrideC;
} MdPic;

static uint32_t info_at(const MdPic *M, int px, int py)
{
    if (px < 0 || py < 0 || px >= (int)M->w || py >= (int)M->h)
        return 0xFFFFFFFEu;
    return M->info[(size_t)(py >> 2) * M->infoPitch + (px >> 2)];
}

/* the luma prediction of one candidate: the job svt_oracle_intra_pu takes, cut from the maps as the reference cuts it from its arrays */
static void md_predict(const MdPic *M, const SvtAmdMdLcu *L, int lcu_x, int lcu_y, const MdStats *st, int mode, uint8_t *pred /* N x N */)
{
    const int N = st->size, x0 = lcu_x + st->x, y0 = lcu_y + st->y;
    SvtAmdIntraPuJob J;
    memset(&J, 0, sizeof(J));
    J.size = (uint32_t)N, J.constrained_intra = 0, J.strong_smoothing = 1; /* GenerateIntraLumaReferenceSamplesMd, EbProductCodingLoop.c:280-295 */
    J.pic_left = L->tile_left && st->x == 0, J.pic_top = L->tile_top && st->y == 0, J.pic_right = L->tile_right && ((st->x + N) & 63) == 0;
    J.bottom_left_ok = (uint8_t)md_bottom_left_ok(st), J.top_right_ok = (uint8_t)md_top_right_ok(st);
    J.luma_mode = (uint8_t)mode, J.chroma_mode = 4;
    for (int k = 0; k < 2 * N / 4; k++) {
        const uint32_t le = info_at(M, x0 - 1, y0 + 4 * k) & 0xFF, te = info_at(M, x0 + 4 * k, y0 - 1) & 0xFF;
        J.mode_left[k] = (uint8_t)le, J.mode_top[k] = (uint8_t)te;
    }
    const uint32_t tl = info_at(M, x0 - 1, y0 - 1) & 0xFF;
    J.mode_tl = (uint8_t)(tl == 0xFE ? 0xFF : tl);
    for (int i = 0; i < 2 * N; i++) {
        const int le = J.mode_left[i >> 2], te = J.mode_top[i >> 2];
        J.left[0][i] = (uint16_t)((le == 0xFE || le == 0xFF) ? 0 : M->rec[(size_t)(y0 + i) * M->pitch + x0 - 1]);
        J.top[0][i] = (uint16_t)((te == 0xFE || te == 0xFF) ? 0 : M->rec[(size_t)(y0 - 1) * M->pitch + x0 + i]);
    }
    J.tl[0] = (uint16_t)((tl == 0xFE || tl == 0xFF) ? 0 : M->rec[(size_t)(y0 - 1) * M->pitch + x0 - 1]);
    svt_oracle_intra_pu(1, &J, pred, (uint32_t)N, NULL, NULL, 0);
}

static void md_neighbors(const MdPic *M, const SvtAmdMdLcu *L, int lcu_x, int lcu_y, const MdStats *st, MdNeighbors *Nb)
{
    const int x0 = lcu_x + st->x, y0 = lcu_y + st->y;
    uint32_t l = info_at(M, x0 - 1, y0), t = info_at(M, x0, y0 - 1);
    if ((L->tile_left && st->x == 0) || (l & 0xFF) == 0xFE)
        l = 0xFFFFFFFFu;
    if ((L->tile_top && st->y == 0) || (t & 0xFF) == 0xFE)
        t = 0xFFFFFFFFu;
    Nb->left_mode = (uint8_t)l, Nb->left_intra = (uint8_t)(l >> 8), Nb->left_depth = (uint8_t)(l >> 16), Nb->left_skip = (uint8_t)(l >> 24);
    Nb->top_mode = (uint8_t)t, Nb->top_intra = (uint8_t)(t >> 8), Nb->top_depth = (uint8_t)(t >> 16), Nb->top_skip = (uint8_t)(t >> 24);
}

static int g_md_debug;

/* IntraPredictionOl (Codec/EbIntraPrediction.c:5427) of the luma block: neighbours = SOURCE samples (UpdateNeighborSamplesArrayOL, :4952:
 * mid-grey beyond the picture, no substitution, no smoothing) */
static void md_predict_ol(const MdPic *M, const MdStats *st, int lcu_x, int lcu_y, int mode, uint8_t *pred)
{
    const int N = st->size, x0 = lcu_x + st->x, y0 = lcu_y + st->y;
    SvtAmdIntraPuJob J;
    memset(&J, 0, sizeof(J));
    J.size = (uint32_t)N, J.bottom_left_ok = J.top_right_ok = 1, J.no_smoothing = 1, J.mode_tl = 2;
    memset(J.mode_left, 2, sizeof(J.mode_left)), memset(J.mode_top, 2, sizeof(J.mode_top));
    J.luma_mode = (uint8_t)mode, J.chroma_mode = 4;
    const uint8_t *src = M->src + (size_t)y0 * M->srcStride + x0;
    for (int i = 0; i < 2 * N; i++)
        J.left[0][i] = J.top[0][i] = 128;
    J.tl[0] = 128;
    if (x0 != 0)
        for (int i = 0; i < 2 * N && y0 + i < (int)M->h; i++)
            J.left[0][i] = src[(ptrdiff_t)i * M->srcStride - 1];
    if (x0 != 0 && y0 != 0)
        J.tl[0] = src[-(ptrdiff_t)M->srcStride - 1];
    if (y0 != 0)
        for (int i = 0; i < 2 * N && x0 + i < (int)M->w; i++)
            J.top[0][i] = src[i - (ptrdiff_t)M->srcStride];
    svt_oracle_intra_pu(1, &J, pred, (uint32_t)N, NULL, NULL, 0);
}

/* IntraPredictionOl of the chroma pair (the chroma mode is always DM, :5530): neighbours = SOURCE chroma samples (UpdateChromaNeighborSamplesArrayOL, :5065) */
static void md_predict_ol_chroma(const MdPic *M, const MdStats *st, int lcu_x, int lcu_y, int mode, uint8_t *cb, uint8_t *cr /* N/2 x N/2 each */)
{
    const int N = st->size, n = N / 2, x0 = (lcu_x + st->x) / 2, y0 = (lcu_y + st->y) / 2, w = (int)M->w / 2, h = (int)M->h / 2;
    SvtAmdIntraPuJob J;
    memset(&J, 0, sizeof(J));
    J.size = (uint32_t)N, J.bottom_left_ok = J.top_right_ok = 1, J.no_smoothing = 1, J.mode_tl = 2;
    memset(J.mode_left, 2, sizeof(J.mode_left)), memset(J.mode_top, 2, sizeof(J.mode_top));
    J.luma_mode = (uint8_t)mode, J.chroma_mode = 4;
    for (int p = 1; p < 3; p++) {
        const uint8_t *src = M->src_c[p - 1] + (size_t)y0 * M->srcStrideC + x0;
        for (int i = 0; i < 2 * n; i++)
            J.left[p][i] = J.top[p][i] = 128;
        J.tl[p] = 128;
        if (x0 != 0)
            for (int i = 0; i < 2 * n && y0 + i < h; i++)
                J.left[p][i] = src[(ptrdiff_t)i * M->srcStrideC - 1];
        if (x0 != 0 && y0 != 0)
            J.tl[p] = src[-(ptrdiff_t)M->srcStrideC - 1];
        if (y0 != 0)
            for (int i = 0; i < 2 * n && x0 + i < w; i++)
                J.top[p][i] = src[i - (ptrdiff_t)M->srcStrideC];
    }
    svt_oracle_intra_pu(1, &J, NULL, 0, cb, cr, (uint32_t)n);
}

/* Inter2Nx2NPuPredictionHevc (Codec/EbInterPrediction.c:468): luma (N x N) and, with cb / cr, the chroma pair (N/2 x N/2 each) */
static void md_predict_inter_c(const MdPic *M, const MdStats *st, int lcu_x, int lcu_y, const MdCand *c, uint8_t *pred, uint8_t *cb, uint8_t *cr)
{
    static __thread uint8_t tcb[32 * 32], tcr[32 * 32], ty[64 * 64];
    SvtAmdInterPuJob J;
    memset(&J, 0, sizeof(J));
    for (int l = 0; l < 2; l++)
        J.mv[l][0] = c->mv[l].x, J.mv[l][1] = c->mv[l].y;
    J.pu_x = (uint16_t)(lcu_x + st->x), J.pu_y = (uint16_t)(lcu_y + st->y), J.pu_w = J.pu_h = st->size, J.pred_dir = c->dir;
    svt_oracle_inter_pu(&J, M->ref[0], M->ref[1], pred ? pred : ty, st->size, cb ? cb : tcb, cr ? cr : tcr, st->size / 2);
}
static void md_predict_inter(const MdPic *M, const MdStats *st, int lcu_x, int lcu_y, const MdCand *c, uint8_t *pred /* N x N */)
{
    md_predict_inter_c(M, st, lcu_x, lcu_y, c, pred, NULL, NULL);
}

/* the spatial neighbours of the candidate lists with the availability GenerateL0L1AmvpMergeLists derives (:2256-2340) */
static void md_inter_neighbors(const MdPic *M, const SvtAmdMdLcu *L, int lcu_x, int lcu_y, const MdStats *st, MdMvUnit nb[5])
{
    const int x0 = lcu_x + st->x, y0 = lcu_y + st->y, N = st->size;
    const int left = L->tile_left && st->x == 0, top = L->tile_top && st->y == 0, right = L->tile_right && ((st->x + N) & 63) == 0;
    const int px[5] = {x0 - 1, x0 - 1, x0 + N, x0 + N - 1, x0 - 1}, py[5] = {y0 + N, y0 + N - 1, y0 - 1, y0 - 1, y0 - 1};
    int ok[5];
    ok[MD_A0] = md_bottom_left_ok(st) && !left;
    ok[MD_A1] = !left;
    ok[MD_B0] = md_top_right_ok(st) && !top && !right;
    ok[MD_B1] = !top;
    ok[MD_B2] = !left && !top;
    for (int k = 0; k < 5; k++) {
        memset(&nb[k], 0, sizeof(nb[k]));
        if (ok[k] && (info_at(M, px[k], py[k]) & 0xFF) == MD_INTER) {
            nb[k] = M->mv[(size_t)(py[k] >> 3) * M->mvPitch + (px[k] >> 3)];
            nb[k].avail = 1;
        }
    }
}

/* ModeDecisionLcu of one LCU against the picture state M (updated).  src: luma source of the PICTURE. */
/* AddChromaEncDec (Codec/EbProductCodingLoop.c:4158-4349) + the merge / skip decision of EncodePass (Codec/EbCodingLoop.c:3838-3882) for a merge unit
 * of the final tree in a CHROMA_MODE_BEST LCU: chroma prediction, chroma full loop, MergeSkipFullCost with the luma terms the mode decision kept */
static int md_ep_unit_kind(const MdPic *M, const SvtAmdMdPicture *P, const SvtAmdMdLcu *L, const SvtAmdCabacCost *cost, const MdCu *u, const MdStats *st,
                           int lcu_x, int lcu_y)
{
    const int N = st->size, Cn = N / 2, x0 = lcu_x + st->x, y0 = lcu_y + st->y;
    static __thread uint8_t py[64 * 64], pc[2][32 * 32];
    static __thread int16_t res[2][32 * 32], q[2][32 * 32], rc[2][32 * 32];
    SvtAmdInterPuJob J;
    memset(&J, 0, sizeof(J));
    for (int l = 0; l < 2; l++)
        J.mv[l][0] = u->mv[l].x, J.mv[l][1] = u->mv[l].y;
    J.pu_x = (uint16_t)x0, J.pu_y = (uint16_t)y0, J.pu_w = J.pu_h = (uint8_t)N, J.pred_dir = u->inter_dir;
    svt_oracle_inter_pu(&J, M->ref[0], M->ref[1], py, (uint32_t)N, pc[0], pc[1], (uint32_t)Cn);
    for (int p = 0; p < 2; p++)
        for (int j = 0; j < Cn; j++)
            for (int i = 0; i < Cn; i++)
                res[p][j * Cn + i] = q[p][j * Cn + i] = (int16_t)(M->src_c[p][(size_t)(y0 / 2 + j) * M->srcStrideC + x0 / 2 + i] - pc[p][j * Cn + i]);
    SvtAmdChromaLoopIn in;
    SvtAmdChromaLoopOut o;
    memset(&in, 0, sizeof(in));
    in.size = (uint32_t)N, in.cb_qp = in.cr_qp = P->chroma_qp, in.slice_type = P->slice_type, in.pf_mode = (uint32_t)md_pf_mode(P), in.cand_type = MD_INTER;
    const int16_t *const rp[2] = {res[0], res[1]};
    int16_t *const qp2[2] = {q[0], q[1]}, *const rcp[2] = {rc[0], rc[1]};
    svt_oracle_full_loop_chroma(cost, &in, rp, qp2, rcp, &o);
    const uint64_t bits[2] = {o.coeff_bits[0], o.coeff_bits[1]}, dist[2][2] = {{o.dist[0][0], o.dist[0][1]}, {o.dist[1][0], o.dist[1][1]}};
    uint64_t mc, sc;
    md_merge_skip_full_cost(P, M->X, u, N, o.cbf, bits, dist, &mc, &sc);
    if (g_md_debug)
        fprintf(stderr, "  ep unit (%d,%d) size %d: cb cbf %u bits %llu dist %llu/%llu cr cbf %u bits %llu dist %llu/%llu -> merge %llu skip %llu\n", st->x, st->y, N, o.cbf[0],
                (unsigned long long)bits[0], (unsigned long long)dist[0][0], (unsigned long long)dist[0][1], o.cbf[1], (unsigned long long)bits[1],
                (unsigned long long)dist[1][0], (unsigned long long)dist[1][1], (unsigned long long)mc, (unsigned long long)sc);
    return md_ep_merge_kind(M->X, L, mc, sc);
}

static void md_lcu(const SvtAmdMdPicture *P, const SvtAmdMdLcu *L, const SvtAmdCabacCost *cost, const uint8_t *src, uint32_t srcStride,
                   const SvtAmdOisLcuResult *ois, int lcu_x, int lcu_y, int lcu_index, MdPic *M, MdLcuState *S, SvtAmdMdLcuOut *out, uint8_t *ep_kind)
{
    /* per depth: the candidate buffers (prediction, reconstructed coefficients) of the unit tested last, and the best candidate's
     * reconstruction (bestCandidateBuffers[depth]->reconPtr), at the unit's position inside the LCU */
    static __thread uint8_t bestRec[4][64 * 64];
    const int lcuH = (int)P->height - lcu_y < 64 ? (int)P->height - lcu_y : 64;
    const int islice = P->slice_type == 2;
    md_construct_cu_array(S, L);
    int cuIdx = 0;
    do {
        int leaf = L->leaf_index[cuIdx];
        MdStats st = md_stats(leaf);
        S->local[leaf].tested = 1;
        S->cu[leaf].split = (uint8_t)((islice && st.depth == 0) ? 1 : L->leaf_split[cuIdx]);
        MdNeighbors Nb;
        md_neighbors(M, L, lcu_x, lcu_y, &st, &Nb);
        md_context_generation(S, leaf, st.y, &Nb);
        S->cu[leaf].split = (uint8_t)md_skip_small_cu(P, L, S, leaf, st.depth);
        uint32_t mpm[3] = {0, 0, 0};
        if (P->mpm_search && !L->restrict_intra_global_motion)
            md_mpm_modes(S->cu[leaf].left_intra_mode, S->cu[leaf].top_intra_mode, mpm);
        const int fullReconSearchCount = md_nfl(P, L, st.size);
        const int N = st.size, x0 = lcu_x + st.x, y0 = lcu_y + st.y;
        /* ProductGenerateAmvpMergeInterIntraMdCandidatesCU (EbModeDecision.c:1795) */
        MdCand cand[MD_MAX_CAND];
        memset(cand, 0, sizeof(cand));
        int ncand = 0;
        if (st.depth != 0 && (islice || st.depth == 3 || !L->restrict_intra_global_motion))
            if (!(P->limit_intra && st.x == 0 && st.y == 0))
                ncand = md_intra_candidates(P, L, ois, leaf, &st, cand);
        if (!islice) {
            MdMvUnit nb[5];
            MdInterLists T;
            memset(&T, 0, sizeof(T));
            md_inter_neighbors(M, L, lcu_x, lcu_y, &st, nb);
            const int totalMerge = md_nmm(P, N);
            md_amvp_merge_lists(P, M->X, nb, M->X->tmvp_enable ? &M->tmvp[lcu_index] : NULL, x0, y0, N, totalMerge, &T);
            ncand = md_inter_candidates(P, L, &M->me[lcu_index].pu[md_raster_index(&st)], &T, (uint32_t)x0, (uint32_t)y0, totalMerge, cand, ncand);
            if (g_md_debug) {
                fprintf(stderr, "  leaf %d amvp L0 %d:(%d,%d)(%d,%d) L1 %d:(%d,%d)(%d,%d) merge %d:", leaf, T.amvp_count[0], T.amvp[0][0].x, T.amvp[0][0].y,
                        T.amvp[0][1].x, T.amvp[0][1].y, T.amvp_count[1], T.amvp[1][0].x, T.amvp[1][0].y, T.amvp[1][1].x, T.amvp[1][1].y, T.merge_count);
                for (int k = 0; k < T.merge_count; k++)
                    fprintf(stderr, " [%d (%d,%d) (%d,%d)]", T.merge[k].dir, T.merge[k].mv[0].x, T.merge[k].mv[0].y, T.merge[k].mv[1].x, T.merge[k].mv[1].y);
                fprintf(stderr, "\n");
            }
        }
        int bufferTotal = fullReconSearchCount;
        ncand = md_mpm_injection(P, L, &st, cand, ncand, &bufferTotal, mpm);
        bufferTotal = ncand < bufferTotal ? ncand : bufferTotal;
        static const int width[4] = {5, 8, 8, 8};
        const int maxBuffers = bufferTotal + 1 < width[st.depth] ? bufferTotal + 1 : width[st.depth];
        /* ProductPerformFastLoop (:1911) */
        uint64_t costs[MD_MAX_CAND], fastLumaRate[MD_MAX_CAND];
        uint8_t evaluated[MD_MAX_CAND];
        static __thread uint8_t pred[64 * 64];
        int bestFirst = -1;
        if (!P->single_fast_loop) { /* first loop: the candidates whose distortion the open-loop stages left (:1948-1988) */
            uint64_t bestCost = ~0ull;
            for (int i = ncand - 1; i >= 0; i--) {
                if (!cand[i].dist_ready)
                    continue;
                uint64_t r;
                const uint64_t c = cand[i].type == MD_INTER ? md_inter_fast_cost(P, &st, &S->cu[leaf], &cand[i], cand[i].me_dist, &r)
                                 : islice               ? md_intra_fast_cost_islice(P, &st, &S->cu[leaf], cand[i].intra_mode, cand[i].me_dist, &r)
                                                        : md_intra_fast_cost_pslice(P, &st, &S->cu[leaf], cand[i].intra_mode, cand[i].me_dist, &r);
                if (c <= bestCost)
                    bestFirst = i, bestCost = c;
            }
        }
        const int chromaFull = L->chroma_encode_mode == 1; /* CHROMA_MODE_FULL: useChromaInformationInFastLoop / InFullLoop (EbModeDecisionProcess.c:439-441) */
        const uint32_t cw = M->X ? M->X->chroma_weight : 0;
        const int Cn = N / 2;
        static __thread uint8_t predC[2][32 * 32];
        for (int i = 0; i < ncand; i++) {
            uint64_t dist = 0, distC = 0;
            evaluated[i] = (uint8_t)(!cand[i].dist_ready || i == bestFirst || P->single_fast_loop);
            costs[i] = ~0ull, fastLumaRate[i] = 0;
            if (!evaluated[i])
                continue;
            const int reuse = i == bestFirst && cand[i].type == MD_INTRA; /* the open-loop distortion stands, no luma prediction (:1660, :2042) */
            if (reuse && P->intra_md_open_loop)
                evaluated[i] = 3;
            if (!cand[i].mpm) {
                if (reuse) {
                    dist = cand[i].me_dist;
                } else {
                    if (cand[i].type == MD_INTER)
                        md_predict_inter(M, &st, lcu_x, lcu_y, &cand[i], pred);
                    else if (P->intra_md_open_loop)
                        md_predict_ol(M, &st, lcu_x, lcu_y, cand[i].intra_mode, pred);
                    else
                        md_predict(M, L, lcu_x, lcu_y, &st, cand[i].intra_mode, pred);
                    dist = svt_oracle_NxMSadKernel(src + (size_t)y0 * srcStride + x0, srcStride, pred, (uint32_t)N, (uint32_t)N, (uint32_t)N);
                }
                if (chromaFull) { /* Cb + Cr SAD of the candidate's chroma prediction (:2053-2077) */
                    if (cand[i].type == MD_INTER)
                        md_predict_inter_c(M, &st, lcu_x, lcu_y, &cand[i], NULL, predC[0], predC[1]);
                    else
                        md_predict_ol_chroma(M, &st, lcu_x, lcu_y, cand[i].intra_mode, predC[0], predC[1]);
                    for (int p = 0; p < 2; p++)
                        distC += svt_oracle_NxMSadKernel(M->src_c[p] + (size_t)(y0 / 2) * M->srcStrideC + x0 / 2, M->srcStrideC, predC[p], (uint32_t)Cn, (uint32_t)Cn, (uint32_t)Cn);
                }
            }
            if (chromaFull)
                distC = md_fast_chroma_noise_rule(L, N, &cand[i], distC);
            costs[i] = cand[i].type == MD_INTER ? md_inter_fast_cost_c(P, &st, &S->cu[leaf], &cand[i], dist, distC, cw, !L->cmplx_noise, &fastLumaRate[i])
                     : islice                 ? md_intra_fast_cost_islice(P, &st, &S->cu[leaf], cand[i].intra_mode, dist, &fastLumaRate[i])
                                              : md_intra_fast_cost_pslice_c(P, &st, &S->cu[leaf], cand[i].intra_mode, dist, distC, cw, &fastLumaRate[i]);
            if (cand[i].mpm)
                costs[i] = 0;
            if (g_md_debug)
                fprintf(stderr, "  leaf %d cand %d type %d mode %d dir %d merge %d/%d mv (%d,%d) (%d,%d) dist %llu fast %llu\n", leaf, i, cand[i].type,
                        cand[i].intra_mode, cand[i].dir, cand[i].merge_flag, cand[i].merge_index, cand[i].mv[0].x, cand[i].mv[0].y, cand[i].mv[1].x,
                        cand[i].mv[1].y, (unsigned long long)dist, (unsigned long long)costs[i]);
        }
        MdBuffers B;
        md_fast_loop_buffers(&B, width[st.depth], maxBuffers, ncand, costs, evaluated);
        bufferTotal = B.evaluated_count < bufferTotal ? B.evaluated_count : bufferTotal;
        uint8_t types[MD_MAX_BUF], best[MD_MAX_BUF];
        for (int b = 0; b < MD_MAX_BUF; b++)
            types[b] = B.cand[b] >= 0 ? cand[B.cand[b]].type : 0;
        const int same = B.evaluated_count == bufferTotal;
        const int fullCount = md_pre_mode_decision(&B, types, same ? bufferTotal : maxBuffers, same, best);
        /* PerformFullLoop (:4351) */
        const int nfull = fullCount < bufferTotal ? fullCount : bufferTotal;
        const int pf = md_pf_mode(P);
        uint32_t ycbf[MD_MAX_BUF] = {0}, fullDist[MD_MAX_BUF] = {0};
        uint64_t mergeCost[MD_MAX_BUF] = {0}, skipCost[MD_MAX_BUF] = {0}, yBits[MD_MAX_BUF] = {0}, yDist[MD_MAX_BUF][2] = {{0, 0}};
        static __thread int16_t reconCoeff[MD_MAX_BUF][64 * 64];
        static __thread uint8_t predBuf[MD_MAX_BUF][64 * 64];
        uint32_t prevRootCbf = 1;
        uint64_t bestFullCost = 0xFFFFFFFFull;
        for (int f = 0; f < nfull; f++) {
            const int b = best[f];
            const MdCand *c = &cand[B.cand[b]];
            if (!islice && c->type == MD_INTRA && prevRootCbf == 0)
                continue;
            /* the buffer's luma prediction: the candidate the fast loop predicted there, or - predictionIsReadyLuma == 0 - a fresh one */
            const int fresh = B.cand[b] == bestFirst && c->type == MD_INTRA && P->intra_md_open_loop && evaluated[B.cand[b]];
            const MdCand *pc = fresh || B.pred[b] < 0 ? c : &cand[B.pred[b]];
            if (pc->type == MD_INTER)
                md_predict_inter(M, &st, lcu_x, lcu_y, pc, predBuf[b]);
            else if (P->intra_md_open_loop)
                md_predict_ol(M, &st, lcu_x, lcu_y, pc->intra_mode, predBuf[b]);
            else
                md_predict(M, L, lcu_x, lcu_y, &st, pc->intra_mode, predBuf[b]);
            static __thread int16_t residual[64 * 64], quant[64 * 64];
            for (int j = 0; j < N; j++)
                for (int i = 0; i < N; i++)
                    residual[j * N + i] = (int16_t)(src[(size_t)(y0 + j) * srcStride + x0 + i] - predBuf[b][j * N + i]);
            memcpy(quant, residual, sizeof(int16_t) * (size_t)(N * N));
            SvtAmdFullLoopIn in;
            SvtAmdFullLoopOut o;
            memset(&in, 0, sizeof(in));
            in.size = (uint32_t)N, in.qp = P->qp, in.slice_type = P->slice_type, in.pf_mode = (uint16_t)pf, in.pm_core = 0, in.cand_type = c->type;
            in.intra_luma_mode = c->intra_mode, in.full_lambda = P->full_lambda;
            in.cbf_bits[0] = P->rates.lumaCbfBits[0], in.cbf_bits[1] = P->rates.lumaCbfBits[1];
            in.cbf_bits[2] = P->rates.lumaCbfBits[5], in.cbf_bits[3] = P->rates.lumaCbfBits[6];
            svt_oracle_product_full_loop_luma(cost, &in, residual, quant, reconCoeff[b], &o);
            ycbf[b] = o.ycbf, fullDist[b] = (uint32_t)o.dist[0];
            const uint64_t bits = L->chroma_encode_mode == 2 /* CHROMA_MODE_BEST */ || chromaFull ? md_pf_coeff_bits(pf, P->qp, o.coeff_bits) : o.coeff_bits;
            yBits[b] = bits, yDist[b][0] = o.dist[0], yDist[b][1] = o.dist[1];
            if (chromaFull) { /* :4443-4560: the candidate's OWN chroma prediction (ChromaPrediction runs when the fast loop did not evaluate it), FullLoop_R +
                               * CuFullDistortionFastTuMode_R, the full cost with chroma */
                static __thread int16_t resC[2][32 * 32], qC[2][32 * 32], rcC[2][32 * 32];
                if (c->type == MD_INTER)
                    md_predict_inter_c(M, &st, lcu_x, lcu_y, c, NULL, predC[0], predC[1]);
                else
                    md_predict_ol_chroma(M, &st, lcu_x, lcu_y, c->intra_mode, predC[0], predC[1]);
                for (int p = 0; p < 2; p++)
                    for (int j = 0; j < Cn; j++)
                        for (int i = 0; i < Cn; i++)
                            resC[p][j * Cn + i] = qC[p][j * Cn + i] = (int16_t)(M->src_c[p][(size_t)(y0 / 2 + j) * M->srcStrideC + x0 / 2 + i] - predC[p][j * Cn + i]);
                SvtAmdChromaLoopIn cin;
                SvtAmdChromaLoopOut co;
                memset(&cin, 0, sizeof(cin));
                cin.size = (uint32_t)N, cin.cb_qp = cin.cr_qp = P->chroma_qp, cin.slice_type = P->slice_type, cin.pf_mode = (uint32_t)pf, cin.cand_type = c->type;
                cin.intra_luma_mode = c->intra_mode;
                const int16_t *const rp[2] = {resC[0], resC[1]};
                int16_t *const qp2[2] = {qC[0], qC[1]}, *const rcp[2] = {rcC[0], rcC[1]};
                svt_oracle_full_loop_chroma(cost, &cin, rp, qp2, rcp, &co);
                const uint64_t cbits[2] = {co.coeff_bits[0], co.coeff_bits[1]}, cdist[2][2] = {{co.dist[0][0], co.dist[0][1]}, {co.dist[1][0], co.dist[1][1]}};
                if (c->type == MD_INTER)
                    B.full_cost[b] = md_inter_full_cost(P, cw, &S->cu[leaf], c, N, o.ycbf, co.cbf, fastLumaRate[B.cand[b]], o.dist, cdist, bits, cbits, &mergeCost[b], &skipCost[b]);
                else
                    B.full_cost[b] = md_intra_full_cost_pslice(P, cw, N, o.ycbf, co.cbf, fastLumaRate[B.cand[b]], o.dist[0], cdist, bits, cbits);
                if (g_md_debug)
                    fprintf(stderr, "    chroma: cb cbf %u bits %llu dist %llu/%llu cr cbf %u bits %llu dist %llu/%llu\n", co.cbf[0], (unsigned long long)cbits[0],
                            (unsigned long long)cdist[0][0], (unsigned long long)cdist[0][1], co.cbf[1], (unsigned long long)cbits[1], (unsigned long long)cdist[1][0],
                            (unsigned long long)cdist[1][1]);
            } else if (c->type == MD_INTER)
                B.full_cost[b] = md_inter_full_luma_cost(P, &S->cu[leaf], c, N, o.ycbf, fastLumaRate[B.cand[b]], o.dist, bits, &mergeCost[b], &skipCost[b]);
            else if (islice)
                B.full_cost[b] = md_intra_full_luma_cost_islice(P, st.lg, o.ycbf, fastLumaRate[B.cand[b]], o.dist[0], bits);
            else
                B.full_cost[b] = md_intra_full_luma_cost_pslice(P, N, o.ycbf, fastLumaRate[B.cand[b]], o.dist[0], bits);
            if (P->full_loop_escape && !islice && c->type == MD_INTER && B.full_cost[b] < bestFullCost)
                prevRootCbf = o.ycbf, bestFullCost = B.full_cost[b];
            if (g_md_debug)
                fprintf(stderr, "  leaf %d full cand %d (type %d mode %d) buf %d: fast %llu ycbf %u dist %llu/%llu bits %llu full %llu\n", leaf, B.cand[b], c->type,
                        c->intra_mode, b, (unsigned long long)B.fast_cost[b], o.ycbf, (unsigned long long)o.dist[0], (unsigned long long)o.dist[1],
                        (unsigned long long)o.coeff_bits, (unsigned long long)B.full_cost[b]);
        }
        /* ProductFullModeDecision (EbModeDecision.c:1995) */
        int lowest = best[0];
        uint64_t lowestCost = ~0ull;
        for (int f = 0; f < fullCount; f++)
            if (B.full_cost[best[f]] < lowestCost)
                lowest = best[f], lowestCost = B.full_cost[best[f]];
        if (ncand > 0) {
            const MdCand *c = &cand[B.cand[lowest]];
            MdCu *u = &S->cu[leaf];
            S->local[leaf].cost = B.full_cost[lowest], S->local[leaf].full_distortion = fullDist[lowest];
            u->pred_mode = c->type, u->skip_flag = 0, u->intra_luma_mode = (uint8_t)(c->type == MD_INTRA ? c->intra_mode : 0x1F);
            u->ycbf = (uint8_t)(N == 64 ? (ycbf[lowest] & 0x1E) : (ycbf[lowest] & 1));
            u->inter_dir = (uint8_t)(c->type == MD_INTER ? c->dir : 3), u->merge_flag = (uint8_t)(c->type == MD_INTER ? c->merge_flag : 0);
            u->merge_index = c->merge_index;
            u->mv[0].x = u->mv[0].y = u->mv[1].x = u->mv[1].y = 0;
            if (c->type == MD_INTER) {
                if (c->dir != MD_L1)
                    u->mv[0] = c->mv[0];
                if (c->dir != MD_L0)
                    u->mv[1] = c->mv[1];
            }
            u->merge_cost = mergeCost[lowest], u->skip_cost = skipCost[lowest];
            u->y_coeff_bits = yBits[lowest], u->y_dist[0] = yDist[lowest][0], u->y_dist[1] = yDist[lowest][1];
            u->fast_luma_rate = fastLumaRate[B.cand[lowest]], u->ycbf_mask = ycbf[lowest];
        }
        if (g_md_debug)
            fprintf(stderr, "leaf %d (%d,%d) size %d: %d candidates, %d buffers, full %d -> type %d mode %d cost %llu\n", leaf, st.x, st.y, st.size, ncand,
                    maxBuffers, nfull, S->cu[leaf].pred_mode, S->cu[leaf].intra_luma_mode, (unsigned long long)S->local[leaf].cost);
        S->local[leaf].mdc_index = (uint8_t)cuIdx;
        int last;
        const int exitParent = md_check_high_cost_partition(P, L, S, leaf);
        if (exitParent >= 0) {
            leaf = exitParent, st = md_stats(leaf), cuIdx = S->local[leaf].mdc_index;
            S->cu[leaf].split = 0;
            last = md_inter_depth_decision(P, S, leaf, lcu_x, lcu_y, 1, 0);
        } else {
            /* PerformInverseTransformRecon (:1334): the best candidate's reconstruction, kept per depth (closed-loop intra only) */
            if (ncand > 0 && !P->intra_md_open_loop && N <= 32) {
                uint8_t *dst = bestRec[st.depth] + st.y * 64 + st.x;
                if (S->cu[leaf].ycbf) {
                    svt_oracle_recon_tu(1, (uint32_t)N, 0, 0, reconCoeff[lowest], predBuf[lowest], (uint32_t)N, dst, 64);
                } else {
                    for (int j = 0; j < N; j++)
                        memcpy(dst + j * 64, predBuf[lowest] + j * N, (size_t)N);
                }
            }
            last = md_inter_depth_decision(P, S, leaf, lcu_x, lcu_y, 0, md_stop_split(P, L, st.depth, S->local[leaf].full_distortion));
        }
        if (S->cu[last].split == 0) { /* ModeDecisionUpdateNeighborArrays (:371) */
            const MdStats ls = md_stats(last);
            const int lx = lcu_x + ls.x, ly = lcu_y + ls.y;
            const MdCu *u = &S->cu[last];
            const uint32_t w = (uint32_t)u->pred_mode | ((uint32_t)u->intra_luma_mode << 8) | ((uint32_t)ls.depth << 16) | ((uint32_t)u->skip_flag << 24);
            MdMvUnit mu;
            memset(&mu, 0, sizeof(mu));
            mu.mv[0] = u->mv[0], mu.mv[1] = u->mv[1], mu.dir = u->inter_dir;
            for (int j = 0; j < ls.size && ly + j < (int)M->h; j++) {
                if (lx < (int)M->w && !P->intra_md_open_loop)
                    memcpy(M->rec + (size_t)(ly + j) * M->pitch + lx, bestRec[ls.depth] + (ls.y + j) * 64 + ls.x,
                           (size_t)(lx + ls.size <= (int)M->w ? ls.size : (int)M->w - lx));
                if ((j & 3) == 0)
                    for (int i = 0; i < ls.size && lx + i < (int)M->w; i += 4)
                        M->info[(size_t)((ly + j) >> 2) * M->infoPitch + ((lx + i) >> 2)] = w;
                if ((j & 7) == 0)
                    for (int i = 0; i < ls.size && lx + i < (int)M->w; i += 8)
                        M->mv[(size_t)((ly + j) >> 3) * M->mvPitch + ((lx + i) >> 3)] = mu;
            }
        }
        if (S->cu[leaf].split)
            cuIdx++;
        else if (lcuH < 64)
            cuIdx++;
        else
            cuIdx += md_next_cu_step(L, cuIdx, st.depth);
    } while (cuIdx < L->leaf_count);
    if (ep_kind) { /* what EncodePass will do with the inter units of the final tree (SVT_AMD_EP_INTER_*; 0xFF: not a final inter unit) */
        memset(ep_kind, 0xFF, SVT_AMD_MD_LEAVES);
        for (int it = 0; it < SVT_AMD_MD_LEAVES;) {
            if (S->cu[it].split) {
                it++;
                continue;
            }
            const MdStats fs = md_stats(it);
            if (lcu_x + fs.x < (int)P->width && lcu_y + fs.y < (int)P->height && S->cu[it].pred_mode == MD_INTER)
                ep_kind[it] = (uint8_t)(!S->cu[it].merge_flag          ? 0
                                        : L->chroma_encode_mode == 1 ? md_ep_merge_kind(M->X, L, S->cu[it].merge_cost, S->cu[it].skip_cost) /* the mode decision's costs hold chroma */
                                                                     : md_ep_unit_kind(M, P, L, cost, &S->cu[it], &fs, lcu_x, lcu_y));
            it += md_depth_offset(fs.depth);
        }
    }
    if (out) {
        memset(out, 0, sizeof(*out));
        for (int i = 0; i < SVT_AMD_MD_LEAVES; i++) {
            const MdCu *u = &S->cu[i];
            out->split[i] = u->split, out->tested[i] = S->local[i].tested, out->pred_mode[i] = u->pred_mode;
            out->intra_luma_mode[i] = u->intra_luma_mode, out->ycbf[i] = u->ycbf, out->cost[i] = S->local[i].cost;
            out->inter_dir[i] = u->inter_dir, out->merge_flag[i] = u->merge_flag, out->merge_index[i] = u->merge_index;
            for (int l = 0; l < 2; l++)
                out->mv[i][l][0] = u->mv[l].x, out->mv[i][l][1] = u->mv[l].y;
            out->merge_cost[i] = u->merge_cost, out->skip_cost[i] = u->skip_cost;
        }
    }
}

/* The mode decision of a whole picture, LCUs in raster order.  src_y: source luma at sample (0,0); lcus / ois / out: one record per LCU.
 * md_rec (optional, width x height, pitch = width): the mode decision's luma reconstruction at the end.  P / B pictures: X, me (one record
 * per LCU), tmvp (the co-located picture's motion field, one record per LCU, or NULL), ref0 / ref1 (HOST planes in the layout of
 * SvtAmdRefPicture).  Returns 0, or -1 when the picture is outside the covered set. */
int svt_oracle_md_picture_inter(const SvtAmdMdPicture *P, const SvtAmdMdInter *X, const SvtAmdMdLcu *lcus, const SvtAmdCabacCost *cost,
                                const uint8_t *src_y, uint32_t stride, const uint8_t *src_cb, const uint8_t *src_cr, uint32_t stride_c,
                                const SvtAmdOisLcuResult *ois, const SvtAmdMeLcuResult *me, const SvtAmdTmvpLcu *tmvp, const SvtAmdRefPicture *ref0,
                                const SvtAmdRefPicture *ref1, SvtAmdMdLcuOut *out, uint8_t *md_rec, uint8_t *ep_kind)
{
    const int wl = (P->width + 63) / 64, hl = (P->height + 63) / 64;
    if (X ? !md_picture_supported_inter(P, X) : !md_picture_supported(P))
        return -1;
    if (X)
        for (int i = 0; i < wl * hl; i++)
            if (!md_lcu_supported(P, &lcus[i]))
                return -1;
    g_md_debug = getenv("SVT_ORACLE_MD_DEBUG") != NULL;
    MdPic M;
    memset(&M, 0, sizeof(M));
    M.w = P->width, M.h = P->height, M.pitch = P->width, M.infoPitch = (uint32_t)(P->width + 3) / 4, M.mvPitch = (uint32_t)(P->width + 7) / 8;
    M.rec = (uint8_t *)calloc((size_t)M.w * M.h, 1);
    M.info = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)M.infoPitch * ((M.h + 3) / 4));
    M.mv = (MdMvUnit *)calloc((size_t)M.mvPitch * ((M.h + 7) / 8), sizeof(MdMvUnit));
    M.X = X, M.me = me, M.tmvp = tmvp, M.ref[0] = ref0, M.ref[1] = ref1, M.src = src_y, M.srcStride = stride;
    M.src_c[0] = src_cb, M.src_c[1] = src_cr, M.srcStrideC = stride_c;
    MdLcuState *S = (MdLcuState *)calloc(1, sizeof(MdLcuState));
    if (!M.rec || !M.info || !M.mv || !S)
        return -2;
    memset(M.info, 0xFF, sizeof(uint32_t) * (size_t)M.infoPitch * ((M.h + 3) / 4));
    for (int ly = 0; ly < hl; ly++)
        for (int lx = 0; lx < wl; lx++) {
            const int i = ly * wl + lx;
            if (g_md_debug)
                fprintf(stderr, "LCU %d (%d,%d)\n", i, lx * 64, ly * 64);
            md_lcu(P, &lcus[i], cost, src_y, stride, &ois[i], lx * 64, ly * 64, i, &M, S, out ? &out[i] : NULL,
                   ep_kind && X && src_cb && src_cr ? ep_kind + (size_t)i * SVT_AMD_MD_LEAVES : NULL);
        }
    if (md_rec)
        memcpy(md_rec, M.rec, (size_t)M.w * M.h);
    free(M.rec), free(M.info), free(M.mv), free(S);
    return 0;
}

int svt_oracle_md_picture(const SvtAmdMdPicture *P, const SvtAmdMdLcu *lcus, const SvtAmdCabacCost *cost, const uint8_t *src_y, uint32_t stride,
                          const SvtAmdOisLcuResult *ois, SvtAmdMdLcuOut *out, uint8_t *md_rec)
{
    return svt_oracle_md_picture_inter(P, NULL, lcus, cost, src_y, stride, NULL, NULL, 0, ois, NULL, NULL, NULL, NULL, out, md_rec, NULL);
}

/* md_mv_bits (md_logic.h) for the table test (tests/test_oracle_md.py::test_mv_bit_table) */
uint32_t svt_oracle_md_mv_bits(int dx, int dy) { return md_mv_bits(dx, dy); }
