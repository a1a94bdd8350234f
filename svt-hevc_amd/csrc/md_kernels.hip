/*
 * Device-resident mode decision + encode pass of whole pictures (include/svt_hevc_amd.h "Device-resident mode decision").
 *
 * Replaces, per LCU, what EncDecKernel's LCU loop runs between ModeDecisionConfigureLcu and the end of EncodePass
 * (Codec/EbEncDecProcess.c:2893-3023): ModeDecisionLcu (Codec/EbProductCodingLoop.c:4691-5114) followed by the LCU's EncodePass
 * (Codec/EbCodingLoop.c:2989; encdec_device.h) on the tree it decided, with the wavefront of AssignEncDecSegments
 * (Codec/EbEncDecProcess.c:1540) over the picture - ONE launch per picture.
 *
 * One 256-thread workgroup per LCU.  The LCU's mode-decision state lives in LDS: the mode decision's own luma reconstruction
 * (mdLumaReconNeighborArray as a picture plane: candidate reconstructions of the units decided so far) with a ring of neighbour samples,
 * the neighbour-array entries of the 4x4 cells (mode type | intra luma mode | depth | skip flag) with the same ring, the source block.
 * The LCU's own inputs (its records, the source block, the picture's controls and rate tables) are loaded BEFORE the workgroup waits for the LCU's neighbours;
 * what the neighbours left in the picture's maps follows the wait.  Per coding unit of the MdcLcuData_t leaf list (I pictures: waves 1..3 idle where P / B pictures
 * use them):
 *   lane 0        context generation, intra candidates                                       (md_logic.h - the text the CPU checker runs)
 *   P / B         beside it, from the unit's first moment: waves 1 and 2 fetch the spatial neighbours' motion (five lanes, a copy per wave) and lane 0 of each makes its
 *                 lists - wave 1 the AMVP candidates of both lists, wave 2 the merge candidates - while wave 3 builds the unit's intra reference (luma, and the chroma
 *                 pair of a CHROMA_MODE_FULL LCU, from SOURCE samples); then a lane per motion-estimation / merge candidate (md_choose_mvp, duplicate check),
 *                 survivors in the scalar order by ballot
 *   lane 0        MPM injection, buffer count;  wave 0: a lane per candidate - first fast loop (best distortion-ready candidate), evaluated flags,
 *                 the packed list of candidates that really need a prediction, prediction slots
 *   wave 0        (I pictures) intra reference of the unit: availability by ballot, substitution, [1 2 1] / strong smoothing   (8.4.4.2.2-3)
 *   4 waves       fast loop: ONE list of tasks (candidate, plane, tile) dealt to the waves - luma blocks (64x64 units: four 32x32 tiles), then, in CHROMA_MODE_FULL LCUs,
 *                 the Cb and Cr blocks of every evaluated candidate; inter prediction through ep_inter_predict_core8 (encdec_device.h: LDS-staged reference windows,
 *                 v_dot4 / v_dot2 filters), intra prediction evaluated per sample in closed form (intra_device.h), SAD by v_sad_u8 on words, added to the candidate's
 *                 LDS accumulator
 *   wave 0        fast costs a lane per candidate (with the chroma distortion and the noise-class rule where the LCU has it), the candidate-buffer replay with the
 *                 buffers in lanes (v_readlane), PreModeDecision
 *   4 waves       full loop: a wave per surviving candidate (64x64 units: a wave per 32x32 transform unit) - residual row per lane, Estimate DCT in
 *                 registers, quantiser, coefficient-domain distortion, coefficient bits (a lane per 4x4 sub-block): the fused unit of the encode pass; CHROMA_MODE_FULL:
 *                 the survivors' chroma pairs (FullLoop_R + CuFullDistortionFastTuMode_R) as tasks on the least loaded waves
 *   wave 0        TuCalcCostLuma + full cost a lane per candidate (InterFullCost / MergeSkipFullCost / IntraFullCostPslice with the chroma terms where the LCU has
 *                 them); lane 0: ProductFullModeDecision, CheckHighCostPartition, (open loop) inter-depth decision
 *   wave 0        (closed loop) the winner's reconstruction (inverse transform + prediction);  lane 0: inter-depth decision
 *   all lanes     neighbour update
 * A 10-bit picture (k_md_encode_picture<INTER, uint16_t>) is decided on the 8-MSB views of its source and reference pictures (MdPictureDev.src / .mref) and encoded
 * on the 10-bit samples (.src16, the picture object's 16-bit planes).
 * then the LCU's final tree becomes an SvtAmdLcuWork record and the encode pass of the LCU runs in the same workgroup.
 * Bounded by latency (an LCU's units are sequential, a picture's wavefront is <= (W/64+1)/2 LCUs wide), not by bytes: algorithmic HBM
 * traffic per LCU = 6 KB source + 6 KB OIS record in, 1.1 KB decisions + the encode pass's 24 KB out.
 */
#include "encdec_device.h"
#include <string.h>
#include <vector>
#include <mutex>
#include <condition_variable>
#define MD_FN __host__ __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
#define MD_LDS(ptr) __builtin_assume(__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void *)(ptr)))
#else
#define MD_LDS(ptr) ((void)0)
#endif
#ifndef MD_RESTAGE_THR
#define MD_RESTAGE_THR 16 /* quarter samples: a neighbour's vector this far from the staged window's centre re-stages the window behind the wait (md_lcu) */
#endif
#ifndef MD_LEAF_CALL
#define MD_LEAF_CALL __noinline__ /* the heavy leaves of the unit chain (interpolation, transform unit) as functions: one copy of their code and registers of their own */
#endif
#include <chrono>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include "md_logic.h"

struct MdPictureDev {
    uint8_t *md_rec;              /* the mode decision's luma reconstruction, sample (0,0) */
    uint32_t md_pitch;
    uint32_t *md_info;            /* per 4x4 luma block: mode type | intra luma mode << 8 | depth << 16 | skip flag << 24; ~0 = never written */
    uint32_t info_pitch;
    const uint8_t *src[3];        /* source planes on the device, sample (0,0) */
    uint32_t src_pitch[2];
    const SvtAmdOisLcuResult *ois;
    const SvtAmdMdLcu *lcus;
    const SvtAmdMdPicture *P;
    SvtAmdMdLcuOut *out;
    /* P / B pictures */
    uint4 *md_mv;                 /* per 8x8 luma block: the unit's MvUnit_t {mv[0], mv[1], direction} (mdMvNeighborArray) */
    uint32_t mv_pitch;
    const SvtAmdMdInter *X;
    const SvtAmdMeLcuResult *me;
    const SvtAmdTmvpLcu *tmvp;
    int encode;                   /* 0: mode decision only (no work record, no encode pass) */
    const SvtAmdCabacCost *cost;  /* the picture's coefficient-rate tables (the picture object's d_cost) */
    /* The mode decision works on 8-bit samples whatever the encoder's bit depth (Inter2Nx2NPuPredictionHevc narrows the 16-bit reference block it reads,
     * UnPackReferenceBlock, Codec/EbInterPrediction.c:414-457; the source is the picture's 8-bit plane): mref = the reference pictures as the mode decision reads them -
     * the picture object's own of an 8-bit picture, their 8-MSB views of a 10-bit one -, src16 = the 10-bit source the encode pass behind it codes (null: 8-bit). */
    EpRefPlanes mref[2];
    const uint16_t *src16[3];
    unsigned long long *prof;     /* debug (svt_amd_debug_md_profile): 16 shader-clock sums per LCU, or null */
    int prof_lcus;                /* LCUs of the picture: the sub-stage sums start behind the stage sums of all of them */
    int force_butterflies;        /* debug (svt_amd_debug_md_force_butterflies): the 16x16 / 32x32 forward transforms of the full loops on the register butterflies (the path a unit
                                   * outside the matrix-core form's wrap-free domain takes) - the tests run the fixtures both ways */
    unsigned long long *trace;    /* debug (-DMD_TRACE builds, svt_amd_debug_md_trace): the time stamps of trace_lcu's units [trace_unit, trace_unit + 2), or null */
    int trace_lcu, trace_unit;
};
/* stage clocks of the mode decision of one LCU, taken by lane 0 behind the barrier that ends the stage.  MD_PROF_ON: `prof_on`, a register copy of "D.prof != null" the
 * functions that use the marks make once per LCU - the descriptor lives in LDS, and fifteen marks per unit each re-reading the pointer were fifteen LDS round trips per unit */
#define MD_PROF_ON prof_on
/* the marks' bodies live in ONE function outside the unit loop: inlined, thirty marks were ~3 KB of instructions strewn over a loop whose code does not fit the
 * instruction cache (a debug facility on the hot path of the product).  q: &M.prof[0]; the fields behind it are laid out as MdShared declares them. */
__device__ __noinline__ void md_prof_mark(unsigned long long *q, int k, int sub)
{
    const unsigned long long c_ = __builtin_readcyclecounter();
    unsigned long long *prof_t = q + 32, *prof_s = q + 33, *prof_d = q + 34;
    const int depth = *reinterpret_cast<const int *>(q + 34 + 128);
    if (sub) {
        q[16 + k] += c_ - *prof_s, prof_d[depth * 32 + 16 + k] += c_ - *prof_s, *prof_s = c_;
    } else {
        q[k] += c_ - *prof_t, prof_d[depth * 32 + k] += c_ - *prof_t, *prof_t = c_, *prof_s = c_;
    }
}
#define MD_PROF(k)                                                          \
    do {                                                                    \
        if (__builtin_expect(MD_PROF_ON, 0) && threadIdx.x == 0)            \
            md_prof_mark(&M.prof[0], k, 0);                                 \
    } while (0)
/* ... and finer marks inside a stage (svt_amd_debug_md_profile_sub): slot k of 16 gets the clocks since the previous mark of either kind */
#define MD_SUB(k)                                                           \
    do {                                                                    \
        if (__builtin_expect(MD_PROF_ON, 0) && threadIdx.x == 0)            \
            md_prof_mark(&M.prof[0], k, 1);                                 \
    } while (0)

/* HASY: the closed-loop decision keeps the mode decision's luma reconstruction of the LCU (+ ring) here; the open-loop one (P / B pictures of this revision) never reads it */
template <bool HASY>
struct MdLocal8T {
    static constexpr int PY = 144, X0 = 16;
    uint8_t y[HASY ? 65 * PY : 16];
    uint32_t info[17 * 36]; /* (cy + 1) * 36 + cx + 1: cy in [-1, 16), cx in [-1, 33] */
    alignas(16) uint8_t src[64 * 64]; /* rows of words: the distortion and residual loops read four samples at a time */
    __device__ __forceinline__ uint8_t *at(int x, int y_) { return &y[(y_ + 1) * PY + X0 + x]; }
    /* neighbour-array entry of the 4x4 cell at luma sample (x, y) relative to the LCU: what lies below the LCU, right of it (from its
     * first row on) or right of the top-right LCU is never written before this LCU */
    __device__ __forceinline__ uint32_t info_at(int x, int y_) const
    {
        const int cx = x >> 2, cy = y_ >> 2;
        if (cy >= 16 || cx >= 32 || (cy >= 0 && cx >= 16))
            return 0xFFFFFFFFu;
        return info[(cy + 1) * 36 + cx + 1];
    }
};

struct MdFl {
    uint32_t nz, d0, d1, bits; /* non-zero levels, sum (coeff - recon)^2, sum coeff^2 over the quantised area, the estimator's bits */
};

/* what only the closed-loop (I picture) / only the inter kernel keeps in LDS */
struct MdClosedLoop {
    alignas(16) uint8_t pred[MD_MAX_BUF][32 * 32];
    int16_t recon_coeff[MD_MAX_BUF][32 * 32];
    uint8_t best_rec[4][64 * 64];
};
#ifdef MD_TRACE
#define MD_PRED_SLOTS 4 /* (the trace buffer needs the room in LDS; a fifth kept candidate is predicted again by the full loop) */
#else
#define MD_PRED_SLOTS 5 /* one motion-estimation candidate + the merge candidates (two or three at the presets' mvMergeSkipModeCount) are evaluated per unit */
#endif
struct MdInterShared {
    /* a wave's chroma prediction of the candidate it works on: the first half of its luma scratch (a wave's tasks follow one another, the luma block is spent by then) */
    __device__ __forceinline__ uint8_t *wpred_c(int wave, int pl) { return wpred[wave] + pl * 1024; }
    MdMvUnit mvu[9 * 18];          /* (cy + 1) * 18 + cx + 1: 8x8 cells, cy in [-1, 8), cx in [-1, 16] */
    SvtAmdMeCuResult me[SVT_AMD_ME_PU_COUNT]; /* the LCU's motion-estimation candidates */
    SvtAmdTmvpLcu tmvp[2];         /* the co-located picture's motion field at this LCU and the one to its right */
    alignas(16) MdCand me_c[4], mg_c[5]; /* the unit's motion-estimation / merge candidates as their list-building waves leave them (wave 0 appends them to the intra candidates) */
    uint32_t nbtab[SVT_AMD_MD_LEAVES][5]; /* per entry of the leaf list, made with the LCU's inputs (off the chain): where its five spatial neighbours A0, A1, B0, B1, B2 lie - index into
                                    * L.info | index into mvu << 10 | (inside what is decided before the unit, not across a tile edge) << 18 */
    uint4 unit_tab[SVT_AMD_MD_LEAVES]; /* per entry of the leaf list: MdStats of its unit (x, y), the unit's index (z): what every thread derives at the top of a unit, made with the LCU's inputs */
    unsigned task_ctr2;            /* ... and of the chroma blocks of the full loop */
    unsigned task_ctr;             /* md_units_inter's fast loop: the next task of the unit's list (the waves draw tasks as they finish: a bi-predicted block costs twice a uni-predicted one) */
    uint2 me_rate[4];              /* ... and the motion-estimation candidates' rate term and fastLumaRate (they depend on the predictors only: derived beside the AMVP lists) */
    int n_me, n_mg;
    alignas(16) uint8_t wpred[4][64 * 64];     /* a wave's prediction of the candidate it works on, pitch = unit size */
    uint8_t cpred[MD_PRED_SLOTS][64 * 64]; /* the fast loop's predictions of the first motion-compensated candidates, kept for the full loop */
    int8_t slot[MD_MAX_CAND];      /* candidate -> cpred slot, -1 = none */
    EpMcScratch<uint8_t> mc[4];
    alignas(16) uint8_t src_c[2][32 * 32];     /* the LCU's chroma source (CHROMA_MODE_FULL candidates; merge / skip decision of the encode pass) */
    /* CHROMA_MODE_FULL LCUs (chroma in both loops of every candidate, EbModeDecisionProcess.c:439-441) */
    alignas(16) uint8_t cpred_c[MD_PRED_SLOTS][2][32 * 32]; /* the chroma predictions beside cpred, pitch = unit size / 2 */
    int16_t refc[2][132];          /* the unit's open-loop chroma intra references (Cb, Cr) in pu_predict's layout */
    uint32_t sadc[MD_MAX_CAND];    /* Cb + Cr SAD of the fast loop */
    uint32_t sadc2[MD_MAX_CAND][2]; /* ... per plane, one owner per entry (md_units_inter) */
    uint8_t heavyc[MD_MAX_CAND];   /* the candidates whose chroma the fast loop predicts and measures, packed */
    int nheavyc;
    MdFl flc[MD_MAX_BUF][2][4];    /* the chroma full loop's sums per buffer, plane and transform unit */
    EpRefPlanes refs[2];           /* the reference pictures' plane pointers and geometry beside the LCU (the kernel argument they come from lives in memory as soon as a
                                    * function takes it by reference: a chain of loads per interpolation otherwise) */
    SvtAmdMdInter X;               /* the picture's inter controls beside the LCU: read per unit (a load from HBM each otherwise) */
    EpRefWindows rw;               /* the luma reference samples around the LCU displaced by the 64x64 unit's motion-estimation vectors, per list (encdec_device.h) */
    EpRefWindowsC rwc;             /* ... and, for CHROMA_MODE_FULL LCUs (chroma in both loops of every candidate), the chroma samples around the same position */
    uint8_t ep_kind[SVT_AMD_MD_LEAVES]; /* SVT_AMD_EP_INTER_* of the final tree's inter units */
    uint8_t fin_leaf[SVT_AMD_LCU_MAX_CUS];
    int nfin;
    __device__ __forceinline__ MdMvUnit *mv_at(int x, int y) { return &mvu[((y >> 3) + 1) * 18 + (x >> 3) + 1]; }
};
template <bool INTER> struct MdVariant { typedef MdClosedLoop type; };
template <> struct MdVariant<true> { typedef MdInterShared type; };

template <bool INTER>
struct MdShared {
    MdLocal8T<!INTER> L;
    MdLcuState S;
    SvtAmdMdLcu lcu;
    SvtAmdOisLcuResult ois;        /* the LCU's open-loop intra search record */
    SvtAmdMdPicture pic;           /* the picture's controls and rate tables: the full costs index the tables by lane-dependent contexts (a load from HBM each otherwise) */
    RateTables rt;                 /* the estimator's scan / context tables (rate_device.h c_rt: 624 B of constant memory read per coefficient position) beside them */
    SvtAmdCabacCost cost;          /* the picture's coefficient-rate tables: read per coefficient in the full loops, so kept beside the LCU instead of in HBM */
    alignas(16) MdCand cand[MD_MAX_CAND];
    unsigned long long costs[MD_MAX_CAND], fast_rate[MD_MAX_CAND];
    uint32_t sad[MD_MAX_CAND];
    uint32_t sadt[MD_MAX_CAND][4]; /* P / B pictures: the fast loop's luma SAD per candidate and tile (one owner per entry; [0] alone for units below 64x64) */
    uint8_t evaluated[MD_MAX_CAND];
    uint8_t heavy[MD_MAX_CAND];    /* the candidates the fast loop has to predict and measure, packed: the waves take them in turn */
    int nheavy;
    MdBuffers B;
    uint8_t types[MD_MAX_BUF], best[MD_MAX_BUF];
    uint32_t ycbf[MD_MAX_BUF];
    MdFl fl[MD_MAX_BUF][4];
    unsigned long long merge_cost[MD_MAX_BUF], skip_cost[MD_MAX_BUF], y_bits[MD_MAX_BUF], y_dist[MD_MAX_BUF][2];
    uint32_t full_dist[MD_MAX_BUF];
    int leaf, cu_idx, ncand, buffer_total, nfull, full_count, max_buffers, lowest, do_recon, exited, last, update, done, best_first, any_intra;
    uint8_t next_step[SVT_AMD_MD_LEAVES + 3]; /* CalculateNextCuIndex's step behind each entry of the leaf list when its unit is not split (md_next_cu_step), made with the LCU's inputs */
    unsigned long long prof[32], prof_t, prof_s;
    unsigned long long prof_d[4][32]; /* the same sums by the depth of the unit they belong to (svt_amd_debug_md_profile_depth) */
    int prof_depth;
    /* (md_prof_mark addresses prof_t .. prof_depth relative to prof[0]) */
    int16_t ref[132], reff[132], border[132];
    typename MdVariant<INTER>::type V;
    int16_t tiles[4][(INTER ? 1 : 2) * TxRegTile<32>::UNIT]; /* a wave's transpose tile (the I picture's inverse transforms run a unit per N lanes: two 32x32 units) */
    int16_t qbuf[4][32 * 32];
};
static_assert(offsetof(MdShared<true>, prof_t) == offsetof(MdShared<true>, prof) + 256 && offsetof(MdShared<true>, prof_s) == offsetof(MdShared<true>, prof) + 264 &&
                  offsetof(MdShared<true>, prof_d) == offsetof(MdShared<true>, prof) + 272 && offsetof(MdShared<true>, prof_depth) == offsetof(MdShared<true>, prof) + 272 + 1024 &&
                  offsetof(MdShared<false>, prof_depth) == offsetof(MdShared<false>, prof) + 272 + 1024,
              "md_prof_mark's view of the profile fields");


/* the unit's intra reference, unfiltered (ref) and filtered (reff), by ONE wave: GenerateLumaIntraReferenceSamplesEncodePass with
 * constrainedIntraFlag 0 / strongIntraSmoothingFlag 1 as GenerateIntraLumaReferenceSamplesMd calls it (Codec/EbProductCodingLoop.c:280-295;
 * Codec/EbIntraPrediction.c:750).  Same scheme as ep_intra_predict_plane (encdec_device.h). */
template <bool INTER>
__device__ __forceinline__ void md_build_refs(MdShared<INTER> &M, const MdStats &st, int lane)
{
    auto &L = M.L;
    const int N = st.size, nb = N >> 2, lgN = st.lg, n = N;
    const bool pic_left = M.lcu.tile_left && st.x == 0, pic_top = M.lcu.tile_top && st.y == 0;
    const bool pic_right = M.lcu.tile_right && ((st.x + N) & 63) == 0;
    const bool bl_ok = md_bottom_left_ok(&st), tr_ok = md_top_right_ok(&st);
    bool a = false;
    if (lane > 4 * nb) {
    } else if (lane < 2 * nb) {
        const int e = (int)(L.info_at(st.x - 1, st.y + 2 * N - 4 - 4 * lane) & 0xFF);
        a = !(e == 0xFE || (!bl_ok && lane < nb) || e == 0xFF || pic_left);
    } else if (lane == 2 * nb) {
        const int e = (int)(L.info_at(st.x - 1, st.y - 1) & 0xFF);
        a = !(e == 0xFE || e == 0xFF || pic_left || pic_top);
    } else {
        const int k = lane - 2 * nb - 1, e = (int)(L.info_at(st.x + 4 * k, st.y - 1) & 0xFF);
        a = !(e == 0xFE || (!tr_ok && k >= nb) || e == 0xFF || pic_top || (pic_right && k >= nb));
    }
    const unsigned long long m = __ballot(a);
    const int firstGroup = m ? __ffsll((long long)m) - 1 : 1 << 30;
    for (int k = lane; k <= 4 * n; k += 64) {
        int v = 128;
        if (firstGroup < (1 << 30)) {
            const int gk = k < 2 * n ? k >> 2 : k == 2 * n ? 2 * nb : 2 * nb + 1 + ((k - 2 * n - 1) >> 2);
            const unsigned long long below = m & ((2ull << gk) - 1ull);
            int src;
            if ((below >> gk) & 1ull) {
                src = k;
            } else if (below) {
                const int sg = 63 - __clzll((long long)below);
                src = sg < 2 * nb ? sg * 4 + 3 : sg == 2 * nb ? 2 * n : 2 * n + 1 + (sg - 2 * nb - 1) * 4 + 3;
            } else {
                src = firstGroup < 2 * nb ? firstGroup * 4 : firstGroup == 2 * nb ? 2 * n : 2 * n + 1 + (firstGroup - 2 * nb - 1) * 4;
            }
            v = src < 2 * n ? (int)*L.at(st.x - 1, st.y + 2 * n - 1 - src) : src == 2 * n ? (int)*L.at(st.x - 1, st.y - 1) : (int)*L.at(st.x + (src - 2 * n - 1), st.y - 1);
        }
        M.border[k] = (int16_t)v;
    }
    EP_WAVE_SYNC();
    const int bl = M.border[0], tlv = M.border[2 * n], tr = M.border[4 * n];
    const bool strong = N >= 32 && abs(bl + tlv - 2 * M.border[n]) < 8 && abs(tlv + tr - 2 * M.border[3 * n]) < 8;
    for (int k = lane; k <= 4 * n; k += 64) {
        const int v = M.border[k];
        int f = v;
        if (strong) {
            if (k > 0 && k < 2 * n)
                f = ((2 * n - k) * bl + k * tlv + n) >> (lgN + 1);
            else if (k > 2 * n && k < 4 * n)
                f = ((2 * n - (k - 2 * n)) * tlv + (k - 2 * n) * tr + n) >> (lgN + 1);
        } else if (k > 0 && k < 4 * n) {
            f = (M.border[k - 1] + 2 * v + M.border[k + 1] + 2) >> 2;
        }
        const int o = k < 2 * n ? 2 * n - 1 - k : k;
        M.ref[o] = (int16_t)v, M.reff[o] = (int16_t)f;
    }
    EP_WAVE_SYNC();
}

/* which reference a luma mode predicts from (intraLumaFilterTable, Codec/EbIntraPrediction.c:60-66) */
__device__ __forceinline__ bool md_mode_filtered(int mode, int lgN)
{
    const int dA = abs(mode - 10), dB = abs(mode - 26), dm = dA < dB ? dA : dB;
    const int thrTab = lgN == 2 ? 35 : lgN == 3 ? 7 : lgN == 4 ? 1 : lgN == 5 ? 0 : 10;
    return dm > thrTab && mode != 1;
}

/* the sum of v over the 64 lanes of a fully active wave, in every lane: data-parallel-primitive adds inside the rows of 16 (quad permutes, half-row and row mirror), the two
 * row broadcasts of gfx9 across them, one v_readlane - no trip through the LDS crossbar (a butterfly of __shfl_xor is six of them in a row) */
__device__ __forceinline__ uint32_t md_wave_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);  /* quad_perm [1,0,3,2] */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);  /* quad_perm [2,3,0,1] */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false); /* row_half_mirror */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false); /* row_mirror: every lane holds its row's sum */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); /* row_bcast15 into rows 1 and 3 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); /* row_bcast31 into rows 2 and 3 */
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int md_dc_value(const int16_t *ref, int n, int lgn, int lane)
{
    int dc = lane < n ? ref[lane] + ref[2 * n + 1 + lane] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        dc += __shfl_xor(dc, o);
    return (dc + n) >> (lgn + 1);
}

/* An intra-predicted block of n x n samples by one wave, FOUR samples a lane and step (pu_predict4: a run along the row for the planar, DC and vertical-class modes, down the
 * column for the horizontal-class ones): stored to dst (pitch n) when dst is given, otherwise measured against src (pitch sp) - the lane's part of the SAD is returned.  One
 * copy of the predictor for the eight places the mode decision predicts an intra block (a sample per lane and step there: ~200 issue slots per four samples, ~70 here). */
__device__ MD_LEAF_CALL uint32_t md_intra_block(int mode, int n, int lgn, const int16_t *use, int luma_edge, int lane, const uint8_t *src, int sp, uint8_t *dst)
{
    MD_LDS(use);
    if (src)
        MD_LDS(src);
    if (dst)
        MD_LDS(dst);
    const int dcv = mode == 1 ? md_dc_value(use, n, lgn, lane) : 0;
    const bool hc = pu_horizontal_class(mode);
    uint32_t sad = 0;
    for (int g = lane; g < (n * n) >> 2; g += 64) {
        const int a = 4 * (g & ((n >> 2) - 1)), b = g >> (lgn - 2);
        const int x = hc ? b : a, y = hc ? a : b;
        int o[4];
        pu_predict4(mode, n, lgn, use, x, y, dcv, luma_edge != 0, 255, o);
        if (!hc) {
            const uint32_t w = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
            if (dst)
                *reinterpret_cast<uint32_t *>(&dst[y * n + x]) = w;
            else
                sad = __builtin_amdgcn_sad_u8(w, *reinterpret_cast<const uint32_t *>(&src[y * sp + x]), sad);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (dst)
                    dst[(y + k) * n + x] = (uint8_t)o[k];
                else
                    sad += (uint32_t)abs(o[k] - (int)src[(y + k) * sp + x]);
            }
        }
    }
    return sad;
}

/* ---- the 16x16 and 32x32 forward transforms of the full loops on the MATRIX CORES (round 6) ----------------------------------------------------------------------------
 * Transform16x16 / Transform32x32Estimate (C_DEFAULT/EbTransforms_C.c: two partial-butterfly passes) are exact integer evaluations of C . X . C^T with a rounding shift
 * after each pass (txfm_mfma.hip; the Estimate forms wrap the first butterfly levels to 16 bits, which no sum reaches while the pass's inputs stay below 2^14 (16x16) /
 * 2^13 (32x32): checked per unit, the register butterflies take the unit otherwise - never on 8-bit video, where |T1| <= 16 . 90 . 255 >> 4).  Here the products run as
 * v_mfma_f32_16x16x16_f16 / v_mfma_f32_32x32x8_f16 ON EXACT INTEGERS: residuals (|x| <= 255) and matrix entries (|c| <= 90) are f16 values, their sums of 16 / 32
 * products stay below 2^24 (exact in the f32 accumulator); the 16-bit intermediate of pass 2 goes in as two exact planes T1 = 64 H + L (|H| <= 512, 0 <= L < 64).
 * Why, on a path that is not FLOP-bound: the unit chain is bound by INSTRUCTIONS ISSUED PER WAVE.  The butterflies keep a row per lane - N lanes of 64 busy, ~560 issue
 * slots for a 16x16 unit and ~800 for its quantiser; the matrix form keeps all 64 lanes busy (4 / 16 coefficients a lane, already where the quantiser wants them) in ~150.
 * Pass 1 is computed transposed (T1^T = S . C^T) so that a lane's accumulator registers ARE its B operand of pass 2 (out = C . T1^T): no transpose, no LDS.
 * The constant operand C (lane l: C_N[l % N][its K slots]) is the same for both passes; a workgroup keeps it in LDS (s_dct_op), filled once per launch. */
typedef _Float16 md_v4h __attribute__((ext_vector_type(4)));
typedef float md_v4f __attribute__((ext_vector_type(4)));
typedef float md_v16f __attribute__((ext_vector_type(16)));
static __shared__ int s_md_force_bfly;       /* debug (MdPictureDev.force_butterflies), set once per launch beside the operands below */
static __shared__ uint32_t s_dct_op[12][64]; /* [0..1]: 16x16, [2 + 2 s .. 3 + 2 s]: 32x32 K-slice s, [10..11]: two 8x8 matrices on the diagonal of a 16x16 one; two dwords = four f16 */
__device__ __forceinline__ void md_dct_operands_init(int force_butterflies)
{
    const int t = threadIdx.x;
    if (t == 64)
        s_md_force_bfly = force_butterflies;
    if (t < 64) {
        union { md_v4h h; uint32_t w[2]; } u;
        for (int i = 0; i < 4; i++)
            u.h[i] = (_Float16)(float)d_T32[2 * (t & 15)][4 * (t >> 4) + i];
        s_dct_op[0][t] = u.w[0], s_dct_op[1][t] = u.w[1];
        for (int i = 0; i < 4; i++) { /* diag(C8, C8): the chroma pair of a 16x16 unit as ONE 16x16 product (md_chroma_pair8) */
            const int r = t & 15, k = 4 * (t >> 4) + i;
            u.h[i] = (_Float16)(float)((r >> 3) == (k >> 3) ? d_T32[4 * (r & 7)][k & 7] : 0);
        }
        s_dct_op[10][t] = u.w[0], s_dct_op[11][t] = u.w[1];
        for (int sl = 0; sl < 4; sl++) {
            for (int i = 0; i < 4; i++)
                u.h[i] = (_Float16)(float)d_T32[t & 31][8 * sl + 4 * (t >> 5) + i];
            s_dct_op[2 + 2 * sl][t] = u.w[0], s_dct_op[3 + 2 * sl][t] = u.w[1];
        }
    }
}
__device__ __forceinline__ md_v4h md_h4(int a, int b, int c, int d)
{
    md_v4h r;
    r[0] = (_Float16)(short)a, r[1] = (_Float16)(short)b, r[2] = (_Float16)(short)c, r[3] = (_Float16)(short)d; /* v_cvt_f16_i16: |values| <= 512, exact */
    return r;
}
__device__ __forceinline__ md_v4h md_dct_op(int slot, int lane)
{
    union { md_v4h h; uint32_t w[2]; } u;
    u.w[0] = s_dct_op[slot][lane], u.w[1] = s_dct_op[slot + 1][lane];
    return u.h;
}
/* coefficients of the N x N unit (N = 16: 4 per lane, coefficient (4 (lane >> 4) + i, lane & 15); N = 32: 16 per lane, coefficient (mfma row(v, lane >> 5), lane & 31)) ->
 * out[]; returns false (in every lane) when the unit leaves the wrap-free domain of the Estimate butterflies */
template <int N>
__device__ __forceinline__ bool md_fwd_mfma(int lane, const uint8_t *src, int srcPitch, const uint8_t *pred, int predPitch, int fs1, int fs2, int (&out)[N * N / 64])
{
    const int off1 = 1 << (fs1 - 1), off2 = 1 << (fs2 - 1);
    if constexpr (N == 16) {
        const int j = lane & 15, g = lane >> 4;
        const uint32_t a = *reinterpret_cast<const uint32_t *>(src + j * srcPitch + 4 * g), b = *reinterpret_cast<const uint32_t *>(pred + j * predPitch + 4 * g);
        const md_v4h A = md_h4((int)(a & 0xFF) - (int)(b & 0xFF), (int)((a >> 8) & 0xFF) - (int)((b >> 8) & 0xFF), (int)((a >> 16) & 0xFF) - (int)((b >> 16) & 0xFF),
                               (int)(a >> 24) - (int)(b >> 24));
        const md_v4h C = md_dct_op(0, lane);
        const md_v4f z = {0.f, 0.f, 0.f, 0.f};
        const md_v4f t1 = __builtin_amdgcn_mfma_f32_16x16x16f16(A, C, z, 0, 0, 0); /* T1^T[4 g + i][lane & 15] */
        int t[4];
        bool wide = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            t[i] = (int)(int16_t)(((int)t1[i] + off1) >> fs1);
            wide = wide || t[i] > 16383 || t[i] < -16383;
        }
        if (__ballot(wide))
            return false;
        const md_v4h H = md_h4(t[0] >> 6, t[1] >> 6, t[2] >> 6, t[3] >> 6), Lo = md_h4(t[0] & 63, t[1] & 63, t[2] & 63, t[3] & 63);
        const md_v4f dh = __builtin_amdgcn_mfma_f32_16x16x16f16(C, H, z, 0, 0, 0), dl = __builtin_amdgcn_mfma_f32_16x16x16f16(C, Lo, z, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++)
            out[i] = (int)(int16_t)((((int)dh[i] << 6) + (int)dl[i] + off2) >> fs2);
        return true;
    } else {
        static_assert(N == 32, "matrix-core forms: 16x16 and 32x32");
        const int m = lane & 31, h = lane >> 5;
        md_v16f acc;
#pragma unroll
        for (int v = 0; v < 16; v++)
            acc[v] = 0.f;
        md_v4h C[4];
#pragma unroll
        for (int sl = 0; sl < 4; sl++) {
            const uint32_t a = *reinterpret_cast<const uint32_t *>(src + m * srcPitch + 8 * sl + 4 * h), b = *reinterpret_cast<const uint32_t *>(pred + m * predPitch + 8 * sl + 4 * h);
            const md_v4h A = md_h4((int)(a & 0xFF) - (int)(b & 0xFF), (int)((a >> 8) & 0xFF) - (int)((b >> 8) & 0xFF), (int)((a >> 16) & 0xFF) - (int)((b >> 16) & 0xFF),
                                   (int)(a >> 24) - (int)(b >> 24));
            C[sl] = md_dct_op(2 + 2 * sl, lane);
            acc = __builtin_amdgcn_mfma_f32_32x32x8f16(A, C[sl], acc, 0, 0, 0);
        }
        int t[16];
        bool wide = false;
#pragma unroll
        for (int v = 0; v < 16; v++) { /* T1^T[row(v, h)][m] */
            t[v] = (int)(int16_t)(((int)acc[v] + off1) >> fs1);
            wide = wide || t[v] > 8191 || t[v] < -8191;
        }
        if (__ballot(wide))
            return false;
        md_v16f dh, dl;
#pragma unroll
        for (int v = 0; v < 16; v++)
            dh[v] = 0.f, dl[v] = 0.f;
#pragma unroll
        for (int sl = 0; sl < 4; sl++) { /* the accumulator registers 4 sl .. 4 sl + 3 hold exactly K-slice sl of pass 2 */
            const md_v4h H = md_h4(t[4 * sl] >> 6, t[4 * sl + 1] >> 6, t[4 * sl + 2] >> 6, t[4 * sl + 3] >> 6);
            const md_v4h Lo = md_h4(t[4 * sl] & 63, t[4 * sl + 1] & 63, t[4 * sl + 2] & 63, t[4 * sl + 3] & 63);
            dh = __builtin_amdgcn_mfma_f32_32x32x8f16(C[sl], H, dh, 0, 0, 0);
            dl = __builtin_amdgcn_mfma_f32_32x32x8f16(C[sl], Lo, dl, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 16; v++)
            out[v] = (int)(int16_t)((((int)dh[v] << 6) + (int)dl[v] + off2) >> fs2);
        return true;
    }
}

/* the coefficient-bit estimator (rate_device.h) as ONE function for every transform size: inlined into the four sizes of md_full_loop_unit it was 2.5 K instructions four
 * times over in a kernel whose unit loop does not fit the instruction cache (SQC_TC_INST_REQ: ~760 instruction lines from L2 per unit, profiles/r06_j) */
__device__ __noinline__ uint32_t md_coeff_bits(const SvtAmdCabacCost *cost, const int16_t *qbuf, int N, int lga, uint32_t nz, int type, int intra_mode, int component, int lane, int S4,
                                               const RateTables *rt)
{
    MD_LDS(cost), MD_LDS(qbuf), MD_LDS(rt);
    const SvtAmdTuInfo ti = {nz, (uint8_t)type, (uint8_t)intra_mode, 4 /* EB_INTRA_CHROMA_DM */, (uint8_t)component};
    return coeff_bits_lanes(*cost, qbuf, (uint32_t)N, lga, ti, lane < S4, lane, lane & (S4 - 1), *rt);
}

/* One transform unit of ProductFullLoop (EbFullLoop.c:185-446) / FullLoop_R + CuFullDistortionFastTuMode_R (:579-1066) on lanes r = 0..N-1 of
 * the calling wave: residual -> EstimateTransform -> quantiser -> coefficient-domain distortion -> coefficient bits.  src / pred: the unit's
 * source and prediction (pitches in samples); recon_coeff: the de-quantised coefficients (N x N, pitch N) for PerformInverseTransformRecon, or
 * null; pf: partial-frequency mode (1 = N2: only the low (N/2)^2 coefficients are quantised, measured and priced); type / component: of the
 * candidate and the plane (rate tables).  Returns (every lane) the unit's sums. */
template <int N>
__device__ MD_LEAF_CALL MdFl md_full_loop_unit(int lane, const uint8_t *src, int srcPitch, const uint8_t *pred, int predPitch, int16_t *recon_coeff, int16_t *tile,
                                                  int16_t *qbuf, int qp, int slice_type, const SvtAmdCabacCost &cost, int type, int intra_mode, int component, int pf, const RateTables &rt)
{
    constexpr int LG = N == 32 ? 5 : N == 16 ? 4 : N == 8 ? 3 : 2;
    constexpr int fs1 = N == 32 ? 6 : N == 16 ? 4 : N == 8 ? 2 : 1, fs2 = N == 4 ? 8 : 9, wrap = N == 32 ? 2 : N == 16 ? 1 : 0;
    MD_LDS(src), MD_LDS(pred), MD_LDS(tile), MD_LDS(qbuf), MD_LDS(&cost), MD_LDS(&rt);
    if (recon_coeff)
        MD_LDS(recon_coeff);
    MD_TR(50);
    const int qpRem = qp % 6, qpPer = qp / 6;
    const uint32_t QF = qpRem == 0 ? 26214u : qpRem == 1 ? 23302u : qpRem == 2 ? 20560u : qpRem == 3 ? 18396u : qpRem == 4 ? 16384u : 14564u;
    const int FFv = qpRem == 0 ? 40 : qpRem == 1 ? 45 : qpRem == 2 ? 51 : qpRem == 3 ? 57 : qpRem == 4 ? 64 : 72;
    const int tshift = 7 - LG, shiftedQBits = 14 + qpPer + tshift;
    const uint32_t offs = ((slice_type == 2 || slice_type == 3) ? 171u : 85u) << (shiftedQBits - 9);
    const int shiftedFFunc = qpPer > 8 ? FFv << (qpPer - 2) : FFv << qpPer;
    const int shiftNum = qpPer > 8 ? 20 - 14 - tshift - 2 : 20 - 14 - tshift, iq_offset = 1 << (shiftNum - 1);
    const int area = N >> pf;
    unsigned nz = 0, d0 = 0, d1 = 0;
    auto quantise = [&](int v, bool inside, int &q, int &c) {
        /* the products of the quantiser fit 24 x 24 bits (coefficients are 16-bit values, the scaling factors below 2^15): v_mul_i32_i24 is a full-rate instruction,
         * v_mul_lo_u32 a quarter-rate one */
        const int sign = v < 0 ? -1 : 1;
        int tq = abs(v);
        tq = (int)__umul24((uint32_t)tq, QF);
        tq = (int)((uint32_t)tq + offs);
        tq >>= shiftedQBits;
        q = clip16i(sign * tq);
        c = clip16i((__mul24(q, shiftedFFunc) + iq_offset) >> shiftNum);
        if (inside) {
            const int df = (int16_t)(v - c);
            nz += q != 0, d0 += (uint32_t)__mul24(df, df), d1 += (uint32_t)__mul24(v, v);
        }
    };
    bool on_matrix_cores = false;
    if constexpr (N == 16 || N == 32) {
        int co[N * N / 64];
        on_matrix_cores = !s_md_force_bfly && md_fwd_mfma<N>(lane, src, srcPitch, pred, predPitch, fs1, fs2, co); /* wave-uniform */
        if (on_matrix_cores) {
            MD_TR(52);
            const int k = lane & (N - 1);
            /* the quantiser without a branch per coefficient (a lane's coefficients lie inside or outside the quantised area one by one: each test was an exec-mask
             * region of its own): a coefficient outside contributes zeros to the sums and lands in a word of the buffer nobody reads */
            const bool col_in = k < area;
#pragma unroll
            for (int i = 0; i < N * N / 64; i++) {
                const int k2 = N == 16 ? 4 * (lane >> 4) + i : (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); /* the accumulator's row */
                const bool inside = col_in && k2 < area;
                const int v = inside ? co[i] : 0;
                int tq = abs(v);
                tq = (int)__umul24((uint32_t)tq, QF);
                tq = (int)((uint32_t)tq + offs);
                tq >>= shiftedQBits;
                const int q = v < 0 ? -tq : tq; /* |q| <= 2^14: no clip */
                const int c = clip16i((__mul24(q, shiftedFFunc) + iq_offset) >> shiftNum);
                const int df = (int16_t)(v - c);
                nz += q != 0, d0 += (uint32_t)__mul24(df, df), d1 += (uint32_t)__mul24(v, v);
                qbuf[k2 * N + k] = (int16_t)q; /* (outside the area: zero, never read) */
                if (recon_coeff && inside)
                    recon_coeff[k2 * N + k] = (int16_t)c;
            }
        }
    }
    if (__builtin_expect(!on_matrix_cores, N < 16)) { /* (16x16 / 32x32: the fall-back of a unit outside the matrix form's domain - out of the way of the hot path's instruction stream) */
    const int r = lane & (N - 1);
    const bool active = lane < N;
    int x[N];
    if (active) { /* rows start on word boundaries (units sit on 4-sample grids of word-aligned planes) */
        const uint32_t *sw = reinterpret_cast<const uint32_t *>(src + r * srcPitch), *pw = reinterpret_cast<const uint32_t *>(pred + r * predPitch);
#pragma unroll
        for (int j = 0; j < N; j += 4) {
            const uint32_t a = sw[j >> 2], b = pw[j >> 2];
#pragma unroll
            for (int k = 0; k < 4; k++)
                x[j + k] = (int)((a >> (8 * k)) & 0xFFu) - (int)((b >> (8 * k)) & 0xFFu);
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; j++)
            x[j] = 0;
    }
    MD_TR(51);
    /* the transform on the unit's N row-lanes (pass 1, transpose through LDS, pass 2), its coefficients straight into the quantiser's buffer (row-major, 16-bit: what the
     * second pass produces) - and the QUANTISER ON ALL 64 LANES: a lane owns N*N/64 consecutive coefficients (16 / 4 / 1 of a 32x32 / 16x16 / 8x8 unit) instead of a whole
     * column on N lanes: a 16x16 unit's quantiser is 4 coefficients deep per lane, not 16 (the unit chain is bound by instructions issued per wave). */
    {
        constexpr int P = TxRegTile<N>::PITCH;
        int16_t *t_ = tile; /* ONE unit per call, on lanes 0..N-1; the other lanes run along on zeros and leave no trace */
        fwd_1d_regs<N>(x, fs1, wrap, [&](int k, int16_t v) {
            if (active)
                t_[k * P + r] = v;
        });
        EP_WAVE_SYNC();
#pragma unroll
        for (int j = 0; j < N; j += 2) {
            const uint32_t w = *(const uint32_t *)&t_[r * P + j];
            x[j] = (int16_t)(w & 0xffffu), x[j + 1] = (int16_t)(w >> 16);
        }
        fwd_1d_regs<N>(x, fs2, wrap, [&](int k, int16_t v) {
            if (active)
                qbuf[k * N + r] = v; /* coefficient (k, r) */
        });
        EP_WAVE_SYNC();
    }
    MD_TR(52);
    {
        constexpr int CPL = N * N >= 64 ? N * N / 64 : 1, LANES = N * N / CPL; /* coefficients per lane; the lanes that hold some */
        const int e0 = lane * CPL;
        const bool lane_in = lane < LANES && (e0 >> LG) < area; /* (a lane's coefficients share their row: CPL <= N) */
        if (lane_in) {
            int v_[CPL];
            int16_t *qp_ = qbuf + e0;
            if constexpr (CPL == 16) {
                const uint4 a = reinterpret_cast<const uint4 *>(qp_)[0], b = reinterpret_cast<const uint4 *>(qp_)[1];
                const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < 8; i++)
                    v_[2 * i] = (int16_t)(w[i] & 0xffffu), v_[2 * i + 1] = (int16_t)(w[i] >> 16);
            } else if constexpr (CPL == 4) {
                const uint2 a = *reinterpret_cast<const uint2 *>(qp_);
                v_[0] = (int16_t)(a.x & 0xffffu), v_[1] = (int16_t)(a.x >> 16), v_[2] = (int16_t)(a.y & 0xffffu), v_[3] = (int16_t)(a.y >> 16);
            } else {
                v_[0] = qp_[0];
            }
            int q_[CPL], c_[CPL];
#pragma unroll
            for (int i = 0; i < CPL; i++) {
                const bool inside = ((e0 + i) & (N - 1)) < area;
                int q, c;
                quantise(v_[i], inside, q, c);
                q_[i] = inside ? q : v_[i] /* outside the quantised area the buffer is never read */, c_[i] = c;
            }
            auto put = [&](int16_t *dst, const int (&val)[CPL]) {
                if constexpr (CPL == 16) {
                    uint32_t w[8];
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        w[i] = (uint32_t)(uint16_t)val[2 * i] | ((uint32_t)(uint16_t)val[2 * i + 1] << 16);
                    reinterpret_cast<uint4 *>(dst)[0] = make_uint4(w[0], w[1], w[2], w[3]), reinterpret_cast<uint4 *>(dst)[1] = make_uint4(w[4], w[5], w[6], w[7]);
                } else if constexpr (CPL == 4) {
                    *reinterpret_cast<uint2 *>(dst) = make_uint2((uint32_t)(uint16_t)val[0] | ((uint32_t)(uint16_t)val[1] << 16), (uint32_t)(uint16_t)val[2] | ((uint32_t)(uint16_t)val[3] << 16));
                } else {
                    dst[0] = (int16_t)val[0];
                }
            };
            put(qp_, q_);
            if (recon_coeff) { /* (outside the quantised area the reconstruction buffer keeps what it held: the closed-loop decision runs without partial frequencies) */
                put(recon_coeff + e0, c_);
            }
        }
    }
    }
    MD_TR(53);
    nz = md_wave_sum(nz), d0 = md_wave_sum(d0), d1 = md_wave_sum(d1); /* the lanes beyond the unit's rows hold zeros */
    EP_WAVE_SYNC(); /* qbuf is written */
    MD_TR(54);
    const int lga = LG - pf, S4 = lga <= 2 ? 1 : 1 << (2 * (lga - 2));
    /* nz is the whole unit's count and the same in every lane: a unit without levels (most merge candidates of a B picture at these QPs) has no bits to estimate */
    const uint32_t b32 = nz ? md_coeff_bits(&cost, qbuf, N, lga, nz, type, intra_mode, component, lane, S4, &rt) : 0u;
    MdFl o;
    o.nz = nz, o.d0 = nz ? d0 : d1, o.d1 = d1, o.bits = nz ? (uint32_t)__shfl((int)b32, 0) : 0u;
    MD_TR(55);
    return o;
}

/* BOTH 8x8 chroma transform units of a 16x16 unit's survivor (FullLoop_R + CuFullDistortionFastTuMode_R, one call per plane in md_chroma_tu) as ONE 16x16 product on the
 * matrix cores: residual and transform matrix block-diagonal - diag(Cb, Cr), diag(C8, C8) - so that C X C^T = diag(C8 Cb C8^T, C8 Cr C8^T).  The register butterflies
 * keep 8 of 64 lanes busy per plane (~1.3 K issue slots each); this form transforms and quantises both planes in ~250.  The 8-point Estimate transform has no 16-bit wrap
 * level: exact for every input, no domain to check (|T1| <= 479 . 255 >> 2 < 2^15; H = T1 >> 6 within +-512, exact in f16).  src / pred: Cb's block, Cr's 1024 bytes
 * behind (the LCU's chroma source planes and the candidates' chroma predictions both lie that way); out: M.V.flc[b][0] (Cr's sums: out[4]).  Same sums, same levels in
 * qbuf (Cb's 64, then Cr's) as two md_chroma_tu calls. */
__device__ MD_LEAF_CALL void md_chroma_pair8(int lane, const uint8_t *src, const uint8_t *pred, int16_t *qbuf, int qp, int slice_type, const SvtAmdCabacCost &cost, int type,
                                             int intra_mode, int pf, const RateTables &rt, MdFl *out)
{
    constexpr int N = 8, LG = 3, fs1 = 2, fs2 = 9;
    MD_LDS(src), MD_LDS(pred), MD_LDS(qbuf), MD_LDS(&cost), MD_LDS(&rt), MD_LDS(out);
    const int pfc = pf == 2 ? 1 : pf; /* correctedPFMode (EbFullLoop.c:647-652) */
    const int qpRem = qp % 6, qpPer = qp / 6;
    const uint32_t QF = qpRem == 0 ? 26214u : qpRem == 1 ? 23302u : qpRem == 2 ? 20560u : qpRem == 3 ? 18396u : qpRem == 4 ? 16384u : 14564u;
    const int FFv = qpRem == 0 ? 40 : qpRem == 1 ? 45 : qpRem == 2 ? 51 : qpRem == 3 ? 57 : qpRem == 4 ? 64 : 72;
    const int tshift = 7 - LG, shiftedQBits = 14 + qpPer + tshift;
    const uint32_t offs = ((slice_type == 2 || slice_type == 3) ? 171u : 85u) << (shiftedQBits - 9);
    const int shiftedFFunc = qpPer > 8 ? FFv << (qpPer - 2) : FFv << qpPer;
    const int shiftNum = qpPer > 8 ? 20 - 14 - tshift - 2 : 20 - 14 - tshift, iq_offset = 1 << (shiftNum - 1);
    const int area = N >> pfc;
    const int j = lane & 15, g = lane >> 4, plane = j >> 3;
    const bool valid = (g >> 1) == plane; /* the lane's four K slots lie in its row's diagonal block */
    const uint8_t *s_ = src + plane * 1024 + (j & 7) * 32 + 4 * (g & 1), *p_ = pred + plane * 1024 + (j & 7) * N + 4 * (g & 1);
    const uint32_t a = valid ? *reinterpret_cast<const uint32_t *>(s_) : 0u, b = valid ? *reinterpret_cast<const uint32_t *>(p_) : 0u;
    const md_v4h A = md_h4((int)(a & 0xFF) - (int)(b & 0xFF), (int)((a >> 8) & 0xFF) - (int)((b >> 8) & 0xFF), (int)((a >> 16) & 0xFF) - (int)((b >> 16) & 0xFF),
                           (int)(a >> 24) - (int)(b >> 24));
    const md_v4h C = md_dct_op(10, lane);
    const md_v4f z = {0.f, 0.f, 0.f, 0.f};
    const md_v4f t1 = __builtin_amdgcn_mfma_f32_16x16x16f16(A, C, z, 0, 0, 0);
    int t[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        t[i] = (int)(int16_t)(((int)t1[i] + (1 << (fs1 - 1))) >> fs1);
    const md_v4h Hh = md_h4(t[0] >> 6, t[1] >> 6, t[2] >> 6, t[3] >> 6), Lo = md_h4(t[0] & 63, t[1] & 63, t[2] & 63, t[3] & 63);
    const md_v4f dh = __builtin_amdgcn_mfma_f32_16x16x16f16(C, Hh, z, 0, 0, 0), dl = __builtin_amdgcn_mfma_f32_16x16x16f16(C, Lo, z, 0, 0, 0);
    unsigned nz = 0, d0 = 0, d1 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int k2 = 4 * g + i; /* coefficient (k2 & 7, j & 7) of plane j >> 3 where the lane is valid */
        const bool inside = valid && (k2 & 7) < area && (j & 7) < area;
        const int v = inside ? (int)(int16_t)((((int)dh[i] << 6) + (int)dl[i] + (1 << (fs2 - 1))) >> fs2) : 0; /* (no branch per coefficient: one outside adds zeros) */
        int tq = abs(v);
        tq = (int)__umul24((uint32_t)tq, QF);
        tq = (int)((uint32_t)tq + offs);
        tq >>= shiftedQBits;
        const int q = v < 0 ? -tq : tq; /* |q| <= 2^14 */
        const int c = clip16i((__mul24(q, shiftedFFunc) + iq_offset) >> shiftNum);
        const int df = (int16_t)(v - c);
        nz += q != 0, d0 += (uint32_t)__mul24(df, df), d1 += (uint32_t)__mul24(v, v);
        if (valid)
            qbuf[plane * 64 + (k2 & 7) * N + (j & 7)] = (int16_t)q;
    }
    /* the sums of each half of the wave (lanes 0..31: Cb's coefficients, 32..63: Cr's) */
    auto halves = [&](unsigned v, unsigned &lo, unsigned &hi) {
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); /* row_bcast15 into rows 1 and 3 */
        lo = (uint32_t)__builtin_amdgcn_readlane((int)v, 31), hi = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    };
    unsigned nzp[2], d0p[2], d1p[2];
    halves(nz, nzp[0], nzp[1]), halves(d0, d0p[0], d0p[1]), halves(d1, d1p[0], d1p[1]);
    EP_WAVE_SYNC(); /* qbuf is written */
    const int lga = LG - pfc, S4 = lga <= 2 ? 1 : 1 << (2 * (lga - 2));
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
        const uint32_t b32 = nzp[pl] ? md_coeff_bits(&cost, qbuf + pl * 64, N, lga, nzp[pl], type, intra_mode, 1 + pl, lane, S4, &rt) : 0u;
        const uint32_t bits = nzp[pl] ? (uint32_t)__shfl((int)b32, 0) : 0u;
        if (lane == 0) { /* md_chroma_tu's scaling of the sums: distortions >> 2 (7 - log2 N), bits << 10 >> 15 */
            constexpr int sh = 2 * (7 - LG);
            MdFl o;
            o.nz = nzp[pl];
            o.d0 = (uint32_t)(((unsigned long long)(nzp[pl] ? d0p[pl] : d1p[pl]) + (1ull << (sh - 1))) >> sh);
            o.d1 = (uint32_t)(((unsigned long long)d1p[pl] + (1ull << (sh - 1))) >> sh);
            o.bits = (uint32_t)((((unsigned long long)bits) << 10) >> 15);
            out[4 * pl] = o;
        }
    }
    EP_WAVE_SYNC();
}

/* PerformInverseTransformRecon of the winner (Codec/EbProductCodingLoop.c:1334-1414): EstimateInvTransform of the de-quantised
 * coefficients + the prediction, clipped, into the per-depth reconstruction at the unit's position (pitch 64) */
template <int N>
__device__ __forceinline__ void md_recon_unit(int lane, const int16_t *recon_coeff, const uint8_t *pred, uint8_t *dst, int16_t *tile)
{
    constexpr int P = TxRegTile<N>::PITCH;
    const int r = lane & (N - 1);
    const bool active = lane < N;
    int16_t *t = tile + (lane / N) * TxRegTile<N>::UNIT;
    int c[N];
#pragma unroll
    for (int j = 0; j < N; j++)
        c[j] = active ? (int)recon_coeff[j * N + r] : 0;
    inv_1d_regs<N>(c, 7, [&](int j, int16_t v) { t[r * P + j] = v; });
    EP_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < N; k++)
        c[k] = t[k * P + r];
    int y[N];
    inv_1d_regs<N>(c, 12, [&](int j, int16_t v) { y[j] = v; });
    if (active) {
#pragma unroll
        for (int j = 0; j < N; j++) {
            const int v = (int)pred[r * N + j] + y[j];
            dst[r * 64 + j] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

/* IntraPredictionOl's reference of the unit (Codec/EbIntraPrediction.c:4952 UpdateNeighborSamplesArrayOL): SOURCE samples around the unit,
 * mid-grey beyond the picture; no substitution, no smoothing.  By one wave; M.ref in pu_predict's layout. */
template <bool INTER>
__device__ __forceinline__ void md_build_refs_ol(const MdPictureDev &D, MdShared<INTER> &M, const MdStats &st, int x0, int y0, int W, int H, int lane)
{
    const int N = st.size;
    const uint8_t *src = D.src[0] + (size_t)y0 * D.src_pitch[0] + x0;
    for (int k = lane; k <= 4 * N; k += 64) {
        int v = 128;
        if (k < 2 * N) {
            if (x0 != 0 && y0 + k < H)
                v = src[(ptrdiff_t)k * D.src_pitch[0] - 1];
        } else if (k == 2 * N) {
            if (x0 != 0 && y0 != 0)
                v = src[-(ptrdiff_t)D.src_pitch[0] - 1];
        } else {
            const int j = k - 2 * N - 1;
            if (y0 != 0 && x0 + j < W)
                v = src[j - (ptrdiff_t)D.src_pitch[0]];
        }
        M.ref[k] = (int16_t)v;
    }
    EP_WAVE_SYNC();
}

/* a 64-bit value of lane l (wave-uniform l): two v_readlane instead of two trips through the LDS crossbar */
__device__ __forceinline__ unsigned long long md_readlane64(unsigned long long v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

/* a window of `rows` rows of `cpr` 8-byte chunks from a reference plane (the core's clamped addressing at the plane's ends) into the wave's scratch, rows WP apart: the
 * fall-back of md_predict_tile for a window outside the staged samples - one copy of the code for every tile size */
__device__ __noinline__ void md_window_from_plane(const uint8_t *plane, int stride, int last, int base0, int rows, int cpr, int lane, uint8_t *win)
{
    constexpr int WP = EpMcScratch<uint8_t>::WP;
    MD_LDS(win);
    const int nchunk = rows * cpr;
    for (int i0 = 0; i0 < nchunk; i0 += 128) { /* two loads in flight per lane */
        uint2 v[2];
        int at[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int i = i0 + 64 * u + lane;
            at[u] = -1;
            if (i < nchunk) {
                const int j = i / cpr, m = i - j * cpr, idx = base0 + j * stride + m * 8;
                at[u] = j * WP + m * 8;
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
        for (int u = 0; u < 2; u++)
            if (at[u] >= 0)
                *reinterpret_cast<uint2 *>(&win[at[u]]) = v[u];
    }
}
/* Inter2Nx2NPuPredictionHevc (Codec/EbInterPrediction.c:468) of ONE TILE (TN x TN samples: the whole block of a unit up to 32x32 / its chroma block, a quarter of a 64x64
 * unit's luma) of plane p of a candidate, by one wave, both lists: ep_inter_predict_core8 (encdec_device.h) with the tile size and the tap count decided at compile time -
 * a function per size (the 16x16 and 8x8 blocks most units have need a fraction of the registers of the 32x32 form: no callee-saved register, so no private-segment traffic
 * around the call), everything passed BY VALUE (a candidate passed by reference went through the private segment: a memory round trip of ~1.5 K clocks per call on the unit
 * chain).  mv0 / mv1: x | y << 16.  td: the tile's first sample in the destination, pitch in samples. */
template <int TN, bool CHROMA>
__device__ MD_LEAF_CALL void md_predict_tile(const EpRefPlanes *refs, int abs_x, int abs_y, int inter_dir, uint32_t mv0, uint32_t mv1, int p, int lane, EpMcScratch<uint8_t> *Mp,
                                             uint8_t *td, int pitch, int tx0, int ty0, const EpRefWindows *RW, const EpRefWindowsC *RWC)
{
    constexpr int WP = EpMcScratch<uint8_t>::WP;
    constexpr int ntaps = CHROMA ? 4 : 8, first = CHROMA ? -1 : -3, rows = TN + ntaps - 1, cpr = (rows + 7) >> 3;
    EpMcScratch<uint8_t> &M = *Mp;
    MD_LDS(Mp), MD_LDS(td), MD_LDS(RW), MD_LDS(RWC), MD_LDS(refs);
    const bool bi = inter_dir == 2;
    bool second = false;
    for (int l = 0; l < 2; l++) {
        if (!(bi || inter_dir == l))
            continue;
        MD_TR(40);
        const EpRefPlanes R = refs[l]; /* one read of the whole record */
        const uint32_t mvw = l ? mv1 : mv0;
        const int mvx = (int)(int16_t)(mvw & 0xFFFF), mvy = (int)(int16_t)(mvw >> 16);
        const int qx = min(max(((abs_x + R.originX) << 2) + mvx, (R.originX - 71) << 2), (R.width + R.originX + 7) << 2);
        const int qy = min(max(((abs_y + R.originY) << 2) + mvy, (R.originY - 71) << 2), (R.height + R.originY + 7) << 2);
        const int ix = (CHROMA ? qx >> 3 : qx >> 2) + tx0, iy = (CHROMA ? qy >> 3 : qy >> 2) + ty0;
        const int fx = __builtin_amdgcn_readfirstlane(CHROMA ? qx & 7 : qx & 3), fy = __builtin_amdgcn_readfirstlane(CHROMA ? qy & 7 : qy & 3);
        MD_TR(41);
        /* where the filter reads the window: a staged one where it lies (EpRefWindows / EpRefWindowsC), any other from the scratch copy the fall-back below makes - ONE
         * instance of the filter per tile size either way (two were ~1 K instructions more per function, in a loop whose code does not fit the instruction cache) */
        /* (the window's place as a byte distance from the scratch copy's: all three live in the workgroup's LDS, 16-byte aligned; an integer, not a choice between pointers) */
        int dbase = 0, doff = 0, dpitch = WP;
        bool staged = false;
        if (!CHROMA) {
            if (RW && RW->valid[l]) {
                const int rx = ix + first - RW->x0[l], ry = iy + first - RW->y0[l];
                if (rx >= 0 && ry >= 0 && rx + rows <= EpRefWindows::P && ry + rows <= EpRefWindows::H)
                    staged = true, dbase = (int)(RW->pix[l] - M.win), doff = ry * EpRefWindows::P + rx, dpitch = EpRefWindows::P;
            }
        } else {
            if (RWC && RWC->valid[l]) {
                const int rx = ix + first - RWC->x0[l], ry = iy + first - RWC->y0[l];
                if (rx >= 0 && ry >= 0 && rx + rows <= EpRefWindowsC::P && ry + rows <= EpRefWindowsC::H)
                    staged = true, dbase = (int)(RWC->pix[l][p - 1] - M.win), doff = ry * EpRefWindowsC::P + rx, dpitch = EpRefWindowsC::P;
            }
        }
        if (__builtin_expect(!staged, 0)) {
            md_window_from_plane((const uint8_t *)R.plane[p], (int)R.stride[CHROMA], R.size[CHROMA] - 1, (iy + first) * (int)R.stride[CHROMA] + ix + first, rows, cpr, lane, M.win);
            EP_WAVE_SYNC();
        }
        MD_TR(42);
        ep_mc8_tile<TN, CHROMA, true>(M, lane, fx, fy, !bi ? 0 : (second ? 2 : 1), td, pitch, M.win + dbase, doff, dpitch);
        MD_TR(44);
        second = true;
    }
}
/* ... of plane p (0 luma: N x N, 1 / 2 chroma: N/2 x N/2) of a candidate (direction, vectors x | y << 16) into dst with pitch = the block's width; tile_first / tile_step
 * let several waves share the four 32x32 tiles of a 64x64 unit's luma */
__device__ __forceinline__ void md_predict_inter_plane(const EpRefPlanes *refs, int dir, uint32_t mv0, uint32_t mv1, int x0, int y0, int N, int p, int lane, EpMcScratch<uint8_t> &mc,
                                                       uint8_t *dst, int tile_first, int tile_step, const EpRefWindows *rw, const EpRefWindowsC *rwc = nullptr)
{
    const int n = p ? N >> 1 : N, pitch = n;
    if (!p) {
        switch (n) {
        case 64:
            for (int ti = tile_first; ti < 4; ti += tile_step) {
                const int ty0 = (ti >> 1) << 5, tx0 = (ti & 1) << 5;
                md_predict_tile<32, false>(refs, x0, y0, dir, mv0, mv1, 0, lane, &mc, dst + ty0 * pitch + tx0, pitch, tx0, ty0, rw, rwc);
            }
            break;
        case 32: md_predict_tile<32, false>(refs, x0, y0, dir, mv0, mv1, 0, lane, &mc, dst, pitch, 0, 0, rw, rwc); break;
        case 16: md_predict_tile<16, false>(refs, x0, y0, dir, mv0, mv1, 0, lane, &mc, dst, pitch, 0, 0, rw, rwc); break;
        default: md_predict_tile<8, false>(refs, x0, y0, dir, mv0, mv1, 0, lane, &mc, dst, pitch, 0, 0, rw, rwc); break;
        }
    } else {
        switch (n) {
        case 32: md_predict_tile<32, true>(refs, x0, y0, dir, mv0, mv1, p, lane, &mc, dst, pitch, 0, 0, rw, rwc); break;
        case 16: md_predict_tile<16, true>(refs, x0, y0, dir, mv0, mv1, p, lane, &mc, dst, pitch, 0, 0, rw, rwc); break;
        case 8: md_predict_tile<8, true>(refs, x0, y0, dir, mv0, mv1, p, lane, &mc, dst, pitch, 0, 0, rw, rwc); break;
        default: md_predict_tile<4, true>(refs, x0, y0, dir, mv0, mv1, p, lane, &mc, dst, pitch, 0, 0, rw, rwc); break;
        }
    }
}
__device__ __forceinline__ uint32_t md_pack_mv(MdMv v) { return (uint32_t)(uint16_t)v.x | ((uint32_t)(uint16_t)v.y << 16); }
__device__ __forceinline__ void md_predict_inter_plane(const EpRefPlanes *refs, const MdCand &c, int x0, int y0, int N, int p, int lane, EpMcScratch<uint8_t> &mc, uint8_t *dst,
                                                       int tile_first, int tile_step, const EpRefWindows *rw, const EpRefWindowsC *rwc = nullptr)
{
    md_predict_inter_plane(refs, (int)c.dir, md_pack_mv(c.mv[0]), md_pack_mv(c.mv[1]), x0, y0, N, p, lane, mc, dst, tile_first, tile_step, rw, rwc);
}
/* IntraPredictionOl's chroma references of the unit (Codec/EbIntraPrediction.c:5065 UpdateChromaNeighborSamplesArrayOL): SOURCE chroma samples around the
 * unit, mid-grey beyond the picture.  By one wave; ref[p] in pu_predict's layout (n = N/2). */
__device__ __forceinline__ void md_build_refs_ol_chroma(const MdPictureDev &D, int16_t (*ref)[132], int N, int x0, int y0, int W, int H, int lane)
{
    const int n = N >> 1, cx = x0 >> 1, cy = y0 >> 1, w = W >> 1, h = H >> 1;
    for (int i = lane; i < 2 * (4 * n + 1); i += 64) {
        const int p = i >= 4 * n + 1, k = p ? i - (4 * n + 1) : i;
        const uint8_t *src = D.src[1 + p] + (size_t)cy * D.src_pitch[1] + cx;
        int v = 128;
        if (k < 2 * n) {
            if (cx != 0 && cy + k < h)
                v = src[(ptrdiff_t)k * D.src_pitch[1] - 1];
        } else if (k == 2 * n) {
            if (cx != 0 && cy != 0)
                v = src[-(ptrdiff_t)D.src_pitch[1] - 1];
        } else {
            const int j = k - 2 * n - 1;
            if (cy != 0 && cx + j < w)
                v = src[j - (ptrdiff_t)D.src_pitch[1]];
        }
        ref[p][k] = (int16_t)v;
    }
    EP_WAVE_SYNC();
}

/* the luma full loop of the unit's candidate on one wave: every transform unit (four 32x32 of a 64x64 unit) -> out[tu] */
__device__ __forceinline__ void md_full_loop_cand(int lane, int N, const uint8_t *src, const uint8_t *pred, int predPitch, int16_t *recon_coeff, int16_t *tile, int16_t *qbuf,
                                                  const SvtAmdMdPicture &P, const SvtAmdCabacCost &cost, const RateTables &rt, int type, int mode, int pf, MdFl *out)
{
    if (N == 64) {
        for (int tu = 0; tu < 4; tu++) {
            const int off = ((tu & 1) << 5) + ((tu >> 1) << 5) * 64, poff = ((tu & 1) << 5) + ((tu >> 1) << 5) * predPitch;
            const MdFl o = md_full_loop_unit<32>(lane, src + off, 64, pred + poff, predPitch, nullptr, tile, qbuf, P.qp, P.slice_type, cost, type, mode, 0, pf, rt);
            if (lane == 0)
                out[tu] = o;
            EP_WAVE_SYNC();
        }
        return;
    }
    MdFl o;
    switch (N) {
    case 32: o = md_full_loop_unit<32>(lane, src, 64, pred, predPitch, recon_coeff, tile, qbuf, P.qp, P.slice_type, cost, type, mode, 0, pf, rt); break;
    case 16: o = md_full_loop_unit<16>(lane, src, 64, pred, predPitch, recon_coeff, tile, qbuf, P.qp, P.slice_type, cost, type, mode, 0, pf, rt); break;
    default: o = md_full_loop_unit<8>(lane, src, 64, pred, predPitch, recon_coeff, tile, qbuf, P.qp, P.slice_type, cost, type, mode, 0, pf, rt); break;
    }
    if (lane == 0)
        out[0] = o;
}

/* TuCalcCostLuma (Codec/EbRateDistortionCost.c:289-375) of one transform unit + the accumulation ProductFullLoop does: *ycbf, *bits, dist[2] */
__device__ __forceinline__ void md_tu_calc_cost(const SvtAmdMdPicture &P, const MdFl &o, int type, int cuSize, int T, int tuIndex, uint32_t *ycbf, unsigned long long *bits,
                                                unsigned long long dist[2])
{
    const int lgT = T == 32 ? 5 : T == 16 ? 4 : 3, dshift = cuSize == 64 ? 4 : 2 * (7 - lgT);
    const unsigned long long d0 = ((unsigned long long)o.d0 + (1ull << (dshift - 1))) >> dshift, d1 = ((unsigned long long)o.d1 + (1ull << (dshift - 1))) >> dshift;
    const unsigned long long tuBits = (((unsigned long long)o.bits) << 10) >> 15;
    const int ctx = cuSize == T;
    const unsigned long long lambda = P.full_lambda;
    const unsigned long long nzRate = (tuBits << 15) + P.rates.lumaCbfBits[5 + ctx], zRate = P.rates.lumaCbfBits[ctx];
    const unsigned long long zCost = type == MD_INTRA ? ~0ull : (d1 << 8) + ((lambda * zRate + (1u << 22)) >> 23);
    const unsigned long long nzCost = (d0 << 8) + ((lambda * nzRate + (1u << 22)) >> 23);
    const bool coded = o.nz != 0 && nzCost < zCost;
    *ycbf |= (uint32_t)coded << tuIndex;
    *bits += nzCost < zCost ? tuBits : 0;
    dist[0] += nzCost < zCost ? d0 : d1, dist[1] += d1;
}

/* one chroma transform unit of FullLoop_R + CuFullDistortionFastTuMode_R on the calling wave -> nz, the two scaled distortions, the bits */
__device__ __forceinline__ void md_chroma_tu(int lane, int T, const uint8_t *src, const uint8_t *pred, int predPitch, int16_t *tile, int16_t *qbuf, const SvtAmdMdPicture &P,
                                             const SvtAmdCabacCost &cost, const RateTables &rt, int type, int mode, int component, int pf, uint32_t *nz, unsigned long long dist[2], unsigned long long *bits)
{
    const int pfc = T == 4 ? 0 : (T == 8 && pf == 2 ? 1 : pf); /* correctedPFMode (EbFullLoop.c:647-652) */
    MdFl o;
    switch (T) {
    case 16: o = md_full_loop_unit<16>(lane, src, 32, pred, predPitch, nullptr, tile, qbuf, P.chroma_qp, P.slice_type, cost, type, mode, component, pfc, rt); break;
    case 8: o = md_full_loop_unit<8>(lane, src, 32, pred, predPitch, nullptr, tile, qbuf, P.chroma_qp, P.slice_type, cost, type, mode, component, pfc, rt); break;
    default: o = md_full_loop_unit<4>(lane, src, 32, pred, predPitch, nullptr, tile, qbuf, P.chroma_qp, P.slice_type, cost, type, mode, component, pfc, rt); break;
    }
    const int lgT = T == 16 ? 4 : T == 8 ? 3 : 2, sh = 2 * (7 - lgT);
    *nz = o.nz;
    dist[0] = ((unsigned long long)o.d0 + (1ull << (sh - 1))) >> sh, dist[1] = ((unsigned long long)o.d1 + (1ull << (sh - 1))) >> sh;
    *bits = (((unsigned long long)o.bits) << 10) >> 15;
    EP_WAVE_SYNC();
}

/* ---- the motion-vector lists of a unit on REGISTERS (round 6): GenerateL0L1AmvpMergeLists (Codec/EbAdaptiveMotionVectorPrediction.c:2117-3005) as md_logic.h restates it
 * (md_amvp_merge_lists_parts - the text the CPU checker runs and the device tests compare with, reference records on both sides), specialised for what a picture fixes:
 * the target picture of list l IS ref_poc[l], so "the neighbour's vector points to the target picture" is a compare of list indices (+ one flag: both lists hold the same
 * picture), and the scale factor of a spatial neighbour that points to the other list's picture is a constant of the picture - the division of ScaleMV happens once per LCU
 * (MdListConsts), not per neighbour.  The five neighbours arrive in registers (v_readlane), both lists are derived side by side: lane l of the wave works on list l. */
struct MdListConsts {
    int same01;      /* ref_poc[0] == ref_poc[1] */
    int need[2];     /* a vector of the OTHER list's picture has to be scaled to reach list l's picture (td != tb) */
    int scale[2];    /* ... by this factor (ScaleMV :28-52) */
    int bslice;
};
__device__ __forceinline__ int md_scale_factor(int tb, int td) /* ScaleMV's factor; tb, td: int16 differences */
{
    tb = md_clip3(-128, 127, tb), td = md_clip3(-128, 127, td);
    const int16_t temp = (int16_t)((0x4000 + ((td >> 1) < 0 ? -(td >> 1) : (td >> 1))) / td);
    return md_clip3(-4096, 4095, (tb * temp + 32) >> 6);
}
__device__ __forceinline__ MdMv md_scale_by(MdMv v, int scale)
{
    MdMv o;
    o.x = (int16_t)md_clip3(-32768, 32767, (scale * v.x + 127 + (scale * v.x < 0)) >> 8);
    o.y = (int16_t)md_clip3(-32768, 32767, (scale * v.y + 127 + (scale * v.y < 0)) >> 8);
    return o;
}
__device__ __forceinline__ MdListConsts md_list_consts(const SvtAmdMdPicture &P, const SvtAmdMdInter &X)
{
    MdListConsts k;
    k.bslice = P.slice_type == 0;
    k.same01 = X.ref_poc[0] == X.ref_poc[1];
    for (int l = 0; l < 2; l++) {
        const int16_t tb = (int16_t)(X.picture_number - X.ref_poc[l]), td = (int16_t)(X.picture_number - X.ref_poc[1 - l]);
        k.need[l] = td != tb;
        k.scale[l] = td != tb && td != 0 ? md_scale_factor(tb, td) : 0;
    }
    return k;
}
/* GetTemporalMVP_V2 / one list of GetTemporalMVPBPicture_V2 (md_temporal_mvp, md_logic.h) */
__device__ __forceinline__ bool md_temporal_mvp_dev(const SvtAmdMdInter &X, const SvtAmdTmvpLcu *map, MdTmvpPos t, int targetList, MdMv *out)
{
    int colList = X.is_low_delay ? targetList : 1 - X.colocated_pu_ref_list;
    const SvtAmdTmvpLcu *m = &map[t.bottom_right ? t.lcu_offset : 0];
    if (!t.bottom_right && !m->available[t.unit])
        return false;
    const int pd = m->pred_dir[t.unit];
    colList = pd == MD_BI ? colList : pd;
    MdMv v;
    v.x = m->mv[colList][t.unit][0], v.y = m->mv[colList][t.unit][1];
    const int16_t td = (int16_t)(X.colocated_poc - m->ref_poc[colList][t.unit]), tb = (int16_t)(X.picture_number - X.ref_poc[targetList]);
    if (td != tb)
        v = md_scale_by(v, md_scale_factor(tb, td));
    *out = v;
    return true;
}
/* A neighbour's motion as THREE PACKED WORDS (mv[0], mv[1]: x | y << 16; dir | avail << 8) and the lists as words too: every selection below is a select between VALUES.
 * (With MdMvUnit records, `scaled(A0.avail ? A0 : A1)` and `i == 0 ? m[0] : m[1]` are selects between ADDRESSES - the compiler then keeps the records in the private segment:
 * 15 scratch stores per unit and list-building wave and memory-latency loads behind them, 130 MB of HBM writes per 4K picture, profiles/r06_aa.) */
/* ... as fifteen VALUE parameters (an aggregate, however it is indexed, invites the optimiser to turn `b0 ? m0[B0] : m0[B1]` back into a load from a selected address) */
#define MD_NB_PARAMS uint32_t A0m0, uint32_t A0m1, uint32_t A0da, uint32_t A1m0, uint32_t A1m1, uint32_t A1da, uint32_t B0m0, uint32_t B0m1, uint32_t B0da, \
                     uint32_t B1m0, uint32_t B1m1, uint32_t B1da, uint32_t B2m0, uint32_t B2m1, uint32_t B2da
#define MD_NB_ARGS(w0, w1, w2) md_rl(w0, MD_A0), md_rl(w1, MD_A0), md_rl(w2, MD_A0), md_rl(w0, MD_A1), md_rl(w1, MD_A1), md_rl(w2, MD_A1), md_rl(w0, MD_B0), md_rl(w1, MD_B0), \
                               md_rl(w2, MD_B0), md_rl(w0, MD_B1), md_rl(w1, MD_B1), md_rl(w2, MD_B1), md_rl(w0, MD_B2), md_rl(w1, MD_B2), md_rl(w2, MD_B2)
__device__ __forceinline__ MdMv md_unpack_mv(uint32_t w)
{
    MdMv v;
    v.x = (int16_t)(w & 0xFFFF), v.y = (int16_t)(w >> 16);
    return v;
}
/* the AMVP candidates of list `list` (a per-lane value): GetSpatialMVPPosAx_V3 / GetNonScalingSpatialMVPPosBx_V3 / GetScalingSpatialMVPPosBx_V3 + the temporal candidate + the
 * zero fill -> c0, c1 (packed), count */
__device__ __forceinline__ int md_amvp_one_list(const MdListConsts &K, const SvtAmdMdInter &X, MD_NB_PARAMS, const SvtAmdTmvpLcu *map, MdTmvpPos tp, int list, uint32_t *c0o, uint32_t *c1o)
{
    const int need = list ? K.need[1] : K.need[0], scale = list ? K.scale[1] : K.scale[0];
    /* non-scaling: the neighbour's vector that points to list `list`'s picture */
    auto nonscale = [&](uint32_t m0, uint32_t m1, uint32_t da, uint32_t *out) -> bool {
        const int dir = (int)(da & 0xFF);
        if (dir == MD_BI) {
            *out = list ? m1 : m0;
            return true;
        }
        const bool ok = dir == list || K.same01;
        if (ok)
            *out = dir ? m1 : m0;
        return ok;
    };
    /* scaling: always available */
    auto scaled = [&](uint32_t m0, uint32_t m1, uint32_t da) -> uint32_t {
        const int dir = (int)(da & 0xFF), l2 = dir == MD_BI ? list : dir;
        uint32_t v = l2 ? m1 : m0;
        if (l2 != list && need)
            v = md_pack_mv(md_scale_by(md_unpack_mv(v), scale));
        return v;
    };
    const bool a0 = (A0da >> 8) & 1, a1 = (A1da >> 8) & 1, b0 = (B0da >> 8) & 1, b1 = (B1da >> 8) & 1, b2 = (B2da >> 8) & 1;
    uint32_t c0 = 0, c1 = 0;
    int num = 0;
    bool ax = false;
    uint32_t v = 0;
    if (a0)
        ax = nonscale(A0m0, A0m1, A0da, &v);
    if (!ax && a1)
        ax = nonscale(A1m0, A1m1, A1da, &v);
    if (!ax && (a0 || a1))
        v = scaled(a0 ? A0m0 : A1m0, a0 ? A0m1 : A1m1, a0 ? A0da : A1da), ax = true;
    if (ax)
        c0 = v, num = 1;
    bool bx = false;
    uint32_t vb = 0;
    if (b0)
        bx = nonscale(B0m0, B0m1, B0da, &vb);
    if (!bx && b1)
        bx = nonscale(B1m0, B1m1, B1da, &vb);
    if (!bx && b2)
        bx = nonscale(B2m0, B2m1, B2da, &vb);
    /* (a vector a failed test left behind is overwritten or never counted, exactly as in the reference's array) */
    if (bx) {
        if (num == 0)
            c0 = vb;
        else
            c1 = vb;
        num++;
    }
    if (!ax && (b0 || b1 || b2)) { /* (ax false: no A neighbour, so at most one candidate so far) */
        const uint32_t vs = scaled(b0 ? B0m0 : b1 ? B1m0 : B2m0, b0 ? B0m1 : b1 ? B1m1 : B2m1,
                                   b0 ? B0da : b1 ? B1da : B2da);
        if (num == 0)
            c0 = vs;
        else
            c1 = vs;
        num++;
    }
    if (num == 2 && c0 == c1)
        num = 1;
    if (map && num < 2) {
        MdMv tv;
        if (md_temporal_mvp_dev(X, map, tp, list, &tv)) {
            if (num == 0)
                c0 = md_pack_mv(tv);
            else
                c1 = md_pack_mv(tv);
            num++;
        }
    }
    if (num < 1 || (num == 1 && (c0 & 0xFFFF) != 0 && (c0 >> 16) != 0)) {
        if (num == 0)
            c0 = 0;
        else
            c1 = 0;
        num++;
    }
    *c0o = c0, *c1o = c1;
    return num;
}
/* ChooseMVPIdx_V2 for one list (md_choose_mvp, md_logic.h): clips the candidate's vector of that list, picks the nearer predictor */
__device__ __forceinline__ void md_choose_mvp_one(const SvtAmdMdPicture &P, uint32_t ox, uint32_t oy, const MdMv a[3], int count, MdMv *mv, uint8_t *idxOut, MdMv *mvp)
{
    md_clip_mv(&P, ox, oy, mv);
    int idx = 0;
    if (count == 2) {
        const uint32_t d0 = (uint32_t)abs(a[0].x - mv->x) + (uint32_t)abs(a[0].y - mv->y), d1 = (uint32_t)abs(a[1].x - mv->x) + (uint32_t)abs(a[1].y - mv->y);
        idx = d0 <= d1 ? 0 : 1;
    } else if (count > 2) {
        return;
    }
    *idxOut = (uint8_t)idx, *mvp = idx ? a[1] : a[0];
}
/* the merge candidates (md_amvp_merge_lists_parts part 4, :2649-2990) as packed words g0[k] / g1[k] / gd[k] (mv[0], mv[1], dir), always totalMerge of them (the zero vectors
 * fill the list).  t0 / t1 / tok: the temporal candidate's two vectors (packed) and whether list 0's exists (derived by the caller, a lane per list) */
struct MdMerge5 {
    uint32_t g0[5], g1[5], gd[5];
};
__device__ __forceinline__ void md_merge_list_regs(const MdListConsts &K, MD_NB_PARAMS, bool have_map, bool tok, uint32_t t0, uint32_t t1, int totalMerge, MdMerge5 &m)
{
    const int bslice = K.bslice;
    int idx = 0;
    auto add = [&](uint32_t dir, uint32_t v0, uint32_t v1) {
        if (idx < totalMerge) { /* (the reference leaves its loop as soon as the list is full) */
#pragma unroll
            for (int k = 0; k < 5; k++)
                if (idx == k)
                    m.g0[k] = v0, m.g1[k] = v1, m.gd[k] = dir;
        }
        idx++;
    };
    auto differs = [&](uint32_t am0, uint32_t am1, uint32_t ada, uint32_t bm0, uint32_t bm1, uint32_t bda) { /* md_mv_differs */
        if (!bslice)
            return am0 != bm0;
        return (ada & 0xFF) != (bda & 0xFF) || am0 != bm0 || am1 != bm1;
    };
    auto dirof = [&](uint32_t da) { return bslice ? (da & 0xFFu) : (uint32_t)MD_L0; };
    const bool a0 = (A0da >> 8) & 1, a1 = (A1da >> 8) & 1, b0 = (B0da >> 8) & 1, b1 = (B1da >> 8) & 1, b2 = (B2da >> 8) & 1;
#pragma unroll
    for (int k = 0; k < 5; k++)
        m.g0[k] = m.g1[k] = m.gd[k] = 0;
    if (a1)
        add(dirof(A1da), A1m0, A1m1);
    if (idx < totalMerge && b1 && (!a1 || differs(B1m0, B1m1, B1da, A1m0, A1m1, A1da)))
        add(dirof(B1da), B1m0, B1m1);
    if (idx < totalMerge && b0 && (!b1 || differs(B0m0, B0m1, B0da, B1m0, B1m1, B1da)))
        add(dirof(B0da), B0m0, B0m1);
    if (idx < totalMerge && a0 && (!a1 || differs(A0m0, A0m1, A0da, A1m0, A1m1, A1da)))
        add(dirof(A0da), A0m0, A0m1);
    if (idx < totalMerge && idx < 4 && b2 && (!a1 || differs(B2m0, B2m1, B2da, A1m0, A1m1, A1da)) && (!b1 || differs(B2m0, B2m1, B2da, B1m0, B1m1, B1da)))
        add(dirof(B2da), B2m0, B2m1);
    if (idx < totalMerge && have_map && tok) {
        if (bslice)
            add(MD_BI, t0, t1);
        else if (idx < 5)
            add(MD_L0, t0, 0u);
    }
    if (idx < totalMerge && bslice) { /* combined bi-predictive candidates: mvMergeCandIndexArrayForFillingUp (:20-23) */
        const int loopEnd = idx * (idx - 1);
        for (int f = 0; idx < totalMerge && f < loopEnd; f++) {
            const int i0 = f == 0 ? 0 : f == 1 ? 1 : f == 2 ? 0 : f == 3 ? 2 : f == 4 ? 1 : f == 5 ? 2 : f == 6 ? 0 : f == 7 ? 3 : f == 8 ? 1 : f == 9 ? 3 : f == 10 ? 2 : 3;
            const int i1 = f == 0 ? 1 : f == 1 ? 0 : f == 2 ? 2 : f == 3 ? 0 : f == 4 ? 2 : f == 5 ? 1 : f == 6 ? 3 : f == 7 ? 0 : f == 8 ? 3 : f == 9 ? 1 : f == 10 ? 3 : 2;
            const uint32_t c0d = i0 == 0 ? m.gd[0] : i0 == 1 ? m.gd[1] : i0 == 2 ? m.gd[2] : m.gd[3], c0v = i0 == 0 ? m.g0[0] : i0 == 1 ? m.g0[1] : i0 == 2 ? m.g0[2] : m.g0[3];
            const uint32_t c1d = i1 == 0 ? m.gd[0] : i1 == 1 ? m.gd[1] : i1 == 2 ? m.gd[2] : m.gd[3], c1v = i1 == 0 ? m.g1[0] : i1 == 1 ? m.g1[1] : i1 == 2 ? m.g1[2] : m.g1[3];
            if (((c0d + 1) & 1) && ((c1d + 1) & 2) && (!K.same01 || c0v != c1v))
                add(MD_BI, c0v, c1v);
        }
    }
    for (int r = 0; r < 5 && idx < totalMerge; r++) /* zero vectors */
        add(bslice ? MD_BI : MD_L0, 0u, 0u);
}

/* ---- the unit loop of P / B pictures (round 6) ---------------------------------------------------------------------------------------------------------------------------
 * The decisions are those of md_lcu's loop below (which I pictures keep) - same md_logic.h rules, same leaves, same arithmetic - arranged for the one thing that bounds the
 * picture: the length of a unit's dependency chain on ONE wave (a wave issues an instruction every ~4 clocks, an LDS round trip costs ~30 issue slots, a workgroup barrier
 * with a single-lane stage behind it serialises four SIMDs).  Four barriers per unit instead of ten:
 *   A  contexts + intra candidates (wave 0) | AMVP lists + motion-estimation candidates (wave 1) | merge list + merge candidates (wave 2) | intra references (wave 3)
 *   -- barrier --
 *   B  EVERY wave derives the unit's candidate list in its own registers (lane i = candidate i: first fast loop, evaluated flags, the task list as ballot masks) - nothing
 *      to broadcast, no barrier - and runs its share of the fast-loop tasks
 *   -- barrier --
 *   C  EVERY wave: fast costs (lane per candidate), candidate-buffer replay, PreModeDecision - again redundantly, so each wave knows the survivors - then its share of the
 *      full-loop tasks
 *   -- barrier --
 *   D  wave 0: full costs (lane per survivor), ProductFullModeDecision, CheckHighCostPartition, inter-depth decision, neighbour update, next unit
 *   -- barrier --
 * Candidates live in lane registers of every wave (a field of candidate c is a v_readlane away), not in LDS records read field by field. */
__device__ __forceinline__ int md_nth_bit(unsigned long long m, int k) /* index of the k-th set bit (k < popcount) */
{
    for (int i = 0; i < k; i++)
        m &= m - 1;
    return __ffsll((long long)m) - 1;
}
__device__ __forceinline__ uint32_t md_rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
union MdCandWords {
    MdCand c;
    uint32_t w[8];
};
static_assert(sizeof(MdCand) == 32, "a candidate is eight words: type | intra_mode | mpm | dist_ready, me_dist, dir | merge_flag | merge_index | mvp_idx[0], mvp_idx[1], mv[0], mv[1], mvp[0], mvp[1]");

/* PROF: whether the stage marks exist at all - the launches of a profiled call (svt_amd_debug_md_profile) take the instance with them, every other launch the one without:
 * thirty tests of a flag per unit and wave, and their branches between the stages, are not on the product's path */
/* CFULL: the LCU is CHROMA_MODE_FULL (chroma in both loops of every candidate) - an instance of the loop per chroma mode, chosen per LCU: the luma-only LCUs' instance carries
 * none of the chroma tasks' code between its stages */
template <bool PROF, bool CFULL>
__device__ __forceinline__ void md_units_inter(const MdPictureDev &D, int lcu, int lcu_x, int lcu_y, MdShared<true> &M)
{
    const bool prof_on = PROF && __builtin_amdgcn_readfirstlane((int)(D.prof != nullptr)) != 0;
    const SvtAmdMdPicture &P = M.pic; /* the rate tables (indexed by contexts): read where they are */
    /* the picture's and the LCU's CONTROLS in registers: a copy of the records' scalar parts, made once per LCU (a control read from LDS is a ~120-clock round trip on
     * a chain whose every stage tests a dozen of them).  Ph / Lh go to the rules that read controls only; the rate tables and the leaf list stay behind P / M.lcu. */
    SvtAmdMdPicture Ph;
    __builtin_memcpy(&Ph, &M.pic, offsetof(SvtAmdMdPicture, rates));
    SvtAmdMdLcu Lh;
    __builtin_memcpy(&Lh.tile_left, &M.lcu.tile_left, sizeof(SvtAmdMdLcu) - offsetof(SvtAmdMdLcu, tile_left));
    Lh.leaf_count = M.lcu.leaf_count;
    /* the wave index as a UNIFORM value (an SGPR): the compiler cannot know that threadIdx.x >> 6 is the same in all lanes of a wave, and makes every `if (wave == ..)` and
     * every task loop a masked vector region otherwise */
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = t & 63;
    auto &L = M.L;
    const int W = (int)Ph.width, H = (int)Ph.height;
    const int lh = min(64, H - lcu_y);
    const SvtAmdOisLcuResult *ois = &M.ois;
    const int pf = md_pf_mode(&Ph);
    /* what the picture and the LCU fix, read once */
    constexpr bool cfull = CFULL; /* CHROMA_MODE_FULL (Lh.chroma_encode_mode == 1): chroma in both loops of every candidate */
    const bool tile_l = Lh.tile_left != 0, tile_t = Lh.tile_top != 0, tile_r = Lh.tile_right != 0;
    const MdListConsts K = md_list_consts(Ph, M.V.X);
    const uint32_t sfb0 = P.rates.splitFlagBits[0], sfb1 = P.rates.splitFlagBits[1], sfb2 = P.rates.splitFlagBits[2]; /* SplitFlagRate of an unsplit unit, by context */
    const bool tmvp_on = M.V.X.tmvp_enable != 0;
    const unsigned long long lanebit = 1ull << lane, below = lanebit - 1ull;
    /* StopSplitCondition's thresholds of the three depths: what the picture and the LCU fix of md_stop_split (its tables are memory on the device: two dependent loads per
     * call on wave 0's serial stretch of every unit otherwise) */
    const uint32_t ss_thr0 = md_stop_split_threshold(&Ph, &Lh, 0), ss_thr1 = md_stop_split_threshold(&Ph, &Lh, 1), ss_thr2 = md_stop_split_threshold(&Ph, &Lh, 2);
    int cuIdx = M.cu_idx;
    for (;;) {
        MD_TR(10);
        /* ---- the unit (every thread alike) ---- */
        const uint4 ur = M.V.unit_tab[cuIdx]; /* (md_stats of the unit is ~40 instructions and a dependent read of the leaf list: tabulated with the LCU's inputs) */
        const int leaf = (int)ur.z;
        MdStats st;
        {
            const uint32_t w_[2] = {ur.x, ur.y};
            __builtin_memcpy(&st, w_, 8);
        }
        const int N = st.size, lgN = st.lg, x0 = lcu_x + st.x, y0 = lcu_y + st.y;
        const int totalMerge = md_nmm(&Ph, N);
        /* ================= A ================= */
        if (wave == 0) {
            if (lane == 0) {
                M.leaf = leaf;
                if (prof_on)
                    M.prof_depth = st.depth;
                M.S.local[leaf].tested = 1;
                M.S.cu[leaf].split = M.lcu.leaf_split[cuIdx];
                uint32_t l = L.info_at(st.x - 1, st.y), tp = L.info_at(st.x, st.y - 1);
                if ((tile_l && st.x == 0) || (l & 0xFF) == 0xFE)
                    l = 0xFFFFFFFFu;
                if ((tile_t && st.y == 0) || (tp & 0xFF) == 0xFE)
                    tp = 0xFFFFFFFFu;
                MdNeighbors Nb;
                Nb.left_mode = (uint8_t)l, Nb.left_intra = (uint8_t)(l >> 8), Nb.left_depth = (uint8_t)(l >> 16), Nb.left_skip = (uint8_t)(l >> 24);
                Nb.top_mode = (uint8_t)tp, Nb.top_intra = (uint8_t)(tp >> 8), Nb.top_depth = (uint8_t)(tp >> 16), Nb.top_skip = (uint8_t)(tp >> 24);
                md_context_generation(&M.S, leaf, st.y, &Nb);
                M.S.cu[leaf].split = (uint8_t)md_skip_small_cu(&Ph, &Lh, &M.S, leaf, st.depth);
                int ncand = 0;
                if (st.depth != 0 && (st.depth == 3 || !Lh.restrict_intra_global_motion))
                    if (!(Ph.limit_intra && st.x == 0 && st.y == 0))
                        ncand = md_intra_candidates(&Ph, &M.lcu, ois, leaf, &st, M.cand);
                M.ncand = ncand; /* the intra candidates; the other waves' follow */
                M.V.task_ctr = 0, M.V.task_ctr2 = 0;
            }
            MD_TR(11);
            MD_SUB(0);
        } else if (wave < 3) {
            /* the five spatial neighbours (A0, A1, B0, B1, B2; availability as GenerateL0L1AmvpMergeLists derives it, :2256-2340): a lane each, then all five in registers */
            uint32_t w0 = 0, w1 = 0, w2 = 0;
            if (lane < 5) { /* (the mode and the vectors are requested together: one LDS round trip behind the table's) */
                const uint32_t e = M.V.nbtab[cuIdx][lane];
                const uint32_t inf = L.info[e & 1023u];
                const uint32_t *q = reinterpret_cast<const uint32_t *>(&M.V.mvu[(e >> 10) & 255u]);
                const uint32_t q0 = q[0], q1 = q[1], q2 = q[2];
                if (((e >> 18) & 1u) && (inf & 0xFF) == MD_INTER)
                    w0 = q0, w1 = q1, w2 = (q2 & 0xFFu) | 0x100u; /* mv[0], mv[1], dir | avail << 8 */
            }
            MD_TR(12);
            const SvtAmdTmvpLcu *map = tmvp_on ? M.V.tmvp : nullptr;
            MdTmvpPos tp;
            tp.bottom_right = 0, tp.lcu_offset = 0, tp.unit = 0, tp.pad = 0;
            if (map)
                tp = md_tmvp_position(&Ph, map, x0, y0, N);
            const int list = lane & 1; /* list-specific work: lane l on list l */
            bool keep = false;
            MdCandWords cw;
            cw.w[0] = MD_INTER, cw.w[1] = cw.w[2] = cw.w[3] = cw.w[4] = cw.w[5] = cw.w[6] = cw.w[7] = 0;
            if (wave == 1) {
                /* both AMVP lists side by side, then Me2Nx2NCandidatesInjection: a lane per motion-estimation candidate */
                /* the unit's motion-estimation record: requested BEFORE the lists are built (it depends on no neighbour), used behind them */
                static_assert(sizeof(SvtAmdMeCuResult) == 24, "six words: vectors, distortions, directions | count");
                const uint32_t *mew = reinterpret_cast<const uint32_t *>(&M.V.me[md_raster_index(&st)]);
                const uint32_t me_v0 = mew[0], me_v1 = mew[1], me_dw = mew[5], me_dist = mew[2 + (lane < 3 ? lane : 0)];
                uint32_t pa0, pa1;
                const int num = md_amvp_one_list(K, M.V.X, MD_NB_ARGS(w0, w1, w2), map, tp, list, &pa0, &pa1);
                MD_TR(13);
                if (lane < 3 && lane < (int)(me_dw >> 24)) {
                    const int dir = (int)((me_dw >> (8 * lane)) & 0xFFu);
                    if (!(dir == MD_BI && Ph.depth_mode == 0 && Lh.lcu_md_mode == 10)) {
                        keep = true;
                        cw.w[0] = MD_INTER | (1u << 24); /* type | dist_ready << 24 */
                        cw.w[1] = me_dist, cw.w[2] = (uint32_t)dir, cw.w[4] = me_v0, cw.w[5] = me_v1;
                    }
                }
#pragma unroll
                for (int l = 0; l < 2; l++) { /* ChooseMVPIdx_V2 with list l's candidates (lane l holds them) */
                    MdMv al[3];
                    const uint32_t q0 = md_rl(pa0, l), q1 = md_rl(pa1, l);
                    const int cnt = (int)md_rl((uint32_t)num, l);
                    al[0] = md_unpack_mv(q0), al[1] = md_unpack_mv(q1), al[2] = al[1];
                    if (keep && (cw.c.dir == MD_BI || cw.c.dir == l))
                        md_choose_mvp_one(Ph, (uint32_t)x0, (uint32_t)y0, al, cnt, &cw.c.mv[l], &cw.c.mvp_idx[l], &cw.c.mvp[l]);
                }
                if (keep) { /* InterFastCost*sliceOpt's rate of a candidate that is not a merge candidate reads no context of the unit: this wave is not the longest of the four */
                    MdCu none;
                    none.skip_ctx = 0;
                    uint64_t r64 = 0;
                    const uint32_t rt32 = (uint32_t)md_inter_fast_cost_c(&Ph, &st, &none, &cw.c, 0, 0, 0, 1, &r64);
                    M.V.me_rate[__popcll(__ballot(keep) & below)] = make_uint2(rt32, (uint32_t)r64);
                }
            } else {
                /* the temporal candidate's two vectors side by side, the merge list, then ProductMergeSkip2Nx2NCandidatesInjection: a lane per merge candidate */
                MdMv tv;
                tv.x = tv.y = 0;
                bool tok = false;
                if (map && (list == 0 || K.bslice)) {
                    tok = md_temporal_mvp_dev(M.V.X, map, tp, list, &tv);
                    if (!tok)
                        tv.x = tv.y = 0;
                }
                const uint32_t ptv = md_pack_mv(tv), q0 = md_rl(ptv, 0), q1 = md_rl(ptv, 1);
                const bool tok0 = md_rl((uint32_t)tok, 0) != 0;
                MdMerge5 mg;
                md_merge_list_regs(K, MD_NB_ARGS(w0, w1, w2), map != nullptr, tok0, q0, q1, totalMerge, mg);
                MD_TR(13);
                const int k = lane;
                if (k < 5 && k < totalMerge) {
                    const uint32_t c0v = k == 0 ? mg.g0[0] : k == 1 ? mg.g0[1] : k == 2 ? mg.g0[2] : k == 3 ? mg.g0[3] : mg.g0[4];
                    const uint32_t c1v = k == 0 ? mg.g1[0] : k == 1 ? mg.g1[1] : k == 2 ? mg.g1[2] : k == 3 ? mg.g1[3] : mg.g1[4];
                    const uint32_t cd = k == 0 ? mg.gd[0] : k == 1 ? mg.gd[1] : k == 2 ? mg.gd[2] : k == 3 ? mg.gd[3] : mg.gd[4];
                    bool dup = false;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const bool f0 = c0v == mg.g0[j];
                        const bool f1 = cd != MD_L0 && c1v == mg.g1[j];
                        const bool same = cd == MD_L0 ? f0 : (cd == MD_L1 ? f1 : (f0 && f1));
                        dup = dup || (j < k && cd == mg.gd[j] && same);
                    }
                    if (!dup) {
                        keep = true;
                        cw.w[2] = cd | (1u << 8) | ((uint32_t)k << 16); /* dir | merge_flag << 8 | merge_index << 16 */
                        cw.w[4] = c0v, cw.w[5] = c1v;
                    }
                }
            }
            const unsigned long long km = __ballot(keep);
            if (keep) {
                uint4 *dst = reinterpret_cast<uint4 *>(&(wave == 1 ? M.V.me_c : M.V.mg_c)[__popcll(km & below)]);
                dst[0] = make_uint4(cw.w[0], cw.w[1], cw.w[2], cw.w[3]), dst[1] = make_uint4(cw.w[4], cw.w[5], cw.w[6], cw.w[7]);
            }
            if (lane == 0)
                (wave == 1 ? M.V.n_me : M.V.n_mg) = __popcll(km);
            MD_TR(16);
        } else if (st.depth != 0) {
            /* the unit's intra reference from SOURCE samples (IntraPredictionOl; only units below 64x64 have intra candidates) */
            md_build_refs_ol(D, M, st, x0, y0, W, H, lane);
            if (cfull)
                md_build_refs_ol_chroma(D, M.V.refc, N, x0, y0, W, H, lane);
            MD_TR(14);
        }
        MD_SUB(2);
        __syncthreads();
        MD_TR(15);
        MD_SUB(3);
        /* ================= B: the candidate list in the lanes of EVERY wave ================= */
        const int ni = M.ncand, nme = M.V.n_me, nmg = M.V.n_mg, nc = ni + nme + nmg;
        const bool in = lane < nc;
        MdCandWords cw;
        {
            const MdCand *src = lane < ni ? &M.cand[lane] : lane < ni + nme ? &M.V.me_c[lane - ni] : &M.V.mg_c[in ? lane - ni - nme : 0];
            const uint4 a = reinterpret_cast<const uint4 *>(src)[0], b = reinterpret_cast<const uint4 *>(src)[1];
            cw.w[0] = a.x & 0xFF00FFFFu /* mpm = 0: no most-probable-mode search in P / B pictures */, cw.w[1] = a.y, cw.w[2] = a.z, cw.w[3] = a.w, cw.w[4] = b.x, cw.w[5] = b.y, cw.w[6] = b.z,
            cw.w[7] = b.w;
        }
        const MdCand &c = cw.c;
        const int ctype = in ? c.type : 0;
        const MdCu cuv = M.S.cu[leaf]; /* the unit's contexts (wave 0 left them before the barrier) */
        int bufferTotal = md_nfl(&Ph, &Lh, N);
        bufferTotal = nc < bufferTotal ? nc : bufferTotal;
        const int width = st.depth == 0 ? 5 : 8, max_buffers = bufferTotal + 1 < width ? bufferTotal + 1 : width;
        const bool any_intra = ni != 0;
        /* the first fast loop (EbProductCodingLoop.c:1948-1988): the best of the candidates whose distortion the open-loop stages left; the reference walks from the last
         * candidate down with <=: the LOWEST index among equal costs */
        /* A candidate's fast cost is (distortion terms) + (lambda * rate + 2^22 >> 23), and the RATE is known before any distortion is: the rules of md_logic.h, called once
         * per candidate with zero distortion, return exactly that rate term (and fastLumaRate).  Fast costs stay below 2^31 (SAD of at most 64 x 64 8-bit samples << 8, a
         * chroma term of the same size, a rate term below 2^18): everything downstream - the first fast loop here, the fast costs, the candidate buffers and PreModeDecision
         * behind the second barrier - runs on 32-bit values. */
        unsigned long long rate = 0;
        uint32_t rterm = 0;
        if (in) {
            if (lane >= ni && lane < ni + nme) { /* (derived beside the AMVP lists) */
                const uint2 pr = M.V.me_rate[lane - ni];
                rterm = pr.x, rate = pr.y;
            } else {
                rterm = (uint32_t)(c.type == MD_INTER ? md_inter_fast_cost_c(&Ph, &st, &cuv, &c, 0, 0, 0, 1, (uint64_t *)&rate)
                                                      : md_intra_fast_cost_pslice_c(&Ph, &st, &cuv, c.intra_mode, 0, 0, 0, (uint64_t *)&rate));
            }
        }
        int bestFirst = -1;
        {
            const bool ready = in && c.dist_ready;
            const uint32_t cost = ready ? (c.me_dist << 8) + rterm : 0xFFFFFFFFu;
            uint32_t m = 0xFFFFFFFFu;
            unsigned long long rm = __ballot(ready);
            while (rm) {
                const int l = __ffsll((long long)rm) - 1;
                rm &= rm - 1;
                const uint32_t v = md_rl(cost, l);
                if (bestFirst < 0 || v < m)
                    m = v, bestFirst = l;
            }
        }
        int evl = (int)(in && (!c.dist_ready || lane == bestFirst));
        if (evl && lane == bestFirst && c.type == MD_INTRA)
            evl = 3; /* the open-loop distortion stands, no luma prediction (:1660, :2042) */
        /* what the fast loop has to predict + measure: the evaluated candidates except the open-loop intra candidate whose distortion stands (:2042) */
        const bool heavy = in && evl && !(lane == bestFirst && c.type == MD_INTRA);
        const unsigned long long hm = __ballot(heavy);
        const bool hc = in && evl && cfull; /* CHROMA_MODE_FULL: the chroma pair of EVERY evaluated candidate */
        const unsigned long long cm = __ballot(hc);
        const unsigned long long qm = __ballot(evl && c.type == MD_INTER); /* the first MD_PRED_SLOTS inter candidates the loop evaluates keep their prediction for the full loop */
        const int qrank = __popcll(qm & below);
        const int slot = (in && evl && c.type == MD_INTER && qrank < MD_PRED_SLOTS) ? qrank : -1;
        if (wave == 0 && in && lane >= ni) { /* the list as one array (wave 0's full costs read a survivor's record from it) */
            uint4 *dst = reinterpret_cast<uint4 *>(&M.cand[lane]);
            dst[0] = make_uint4(cw.w[0], cw.w[1], cw.w[2], cw.w[3]), dst[1] = make_uint4(cw.w[4], cw.w[5], cw.w[6], cw.w[7]);
        }
        MD_TR(18);
        MD_SUB(6);
        MD_PROF(1);
        MD_PROF(2);
        if (prof_on && t == 0)
            M.prof[13] += (unsigned long long)nc, M.prof[14] += 1, M.prof_d[M.prof_depth][13] += (unsigned long long)nc, M.prof_d[M.prof_depth][14] += 1;
        /* ---- fast loop (ProductPerformFastLoop's second loop): ONE list of tasks = (candidate, plane, tile) dealt to the four waves ---- */
        const bool tiled64 = N == 64 && !any_intra;
        {
            const int nheavy = __popcll(hm), nhc = __popcll(cm);
            const int nl = tiled64 ? nheavy * 4 : nheavy, ntask = nl + 2 * nhc;
            /* up to four tasks: a wave each; more (CHROMA_MODE_FULL LCUs: eight to twelve): the waves DRAW them from a counter as they finish - the tasks differ by a factor of
             * four (a bi-predicted luma block against a uni-predicted 4x4 chroma block), dealt round-robin the slowest wave set the stage's time */
            const bool draw = ntask > 4;
            for (int tk = wave;;) {
                if (draw) {
                    unsigned got = 0;
                    if (lane == 0)
                        got = atomicAdd(&M.V.task_ctr, 1u);
                    tk = (int)md_rl(got, 0);
                }
                if (tk >= ntask)
                    break;
                MD_TR(20);
                const bool luma = tk < nl;
                /* drawn tasks run from the LAST candidate to the first: the list holds the intra candidates first, and their blocks are the short tasks - the long ones
                 * (motion-compensated, two lists) start first, the short ones fill the waves' tails */
                const int kq = luma ? (tiled64 ? tk >> 2 : tk) : (tk - nl) >> 1, ti = luma ? (tiled64 ? tk & 3 : 0) : 0, pl = luma ? 0 : 1 + ((tk - nl) & 1);
                const int k = draw ? (luma ? nheavy : nhc) - 1 - kq : kq;
                const int ci = md_nth_bit(luma ? hm : cm, k);
                const uint32_t cw0 = md_rl(cw.w[0], ci), cw2 = md_rl(cw.w[2], ci);
                uint32_t sad = 0;
                if ((cw0 & 0xFF) == MD_INTER) {
                    const int sl = (int)md_rl((uint32_t)slot, ci), n = luma ? N : N >> 1, lgn = luma ? lgN : lgN - 1;
                    uint8_t *pr = luma ? (sl >= 0 ? M.V.cpred[sl] : M.V.wpred[wave]) : (sl >= 0 ? M.V.cpred_c[sl][pl - 1] : M.V.wpred_c(wave, pl - 1));
                    MD_TR(21);
                    md_predict_inter_plane(M.V.refs, (int)(cw2 & 0xFF), md_rl(cw.w[4], ci), md_rl(cw.w[5], ci), x0, y0, N, pl, lane, M.V.mc[wave], pr, tiled64 && luma ? ti : 0,
                                           tiled64 && luma ? 4 : 1, &M.V.rw, &M.V.rwc);
                    MD_TR(22);
                    MD_SUB(7);
                    if (luma && tiled64) {
                        const int ty0 = (ti >> 1) << 5, tx0 = (ti & 1) << 5;
                        for (int e = 4 * lane; e < 32 * 32; e += 256) { /* v_sad_u8: four samples a word */
                            const int y = ty0 + (e >> 5), x = tx0 + (e & 31);
                            sad = __builtin_amdgcn_sad_u8(*reinterpret_cast<const uint32_t *>(&pr[y * 64 + x]), *reinterpret_cast<const uint32_t *>(&L.src[(st.y + y) * 64 + st.x + x]), sad);
                        }
                    } else if (luma) {
                        for (int e = 4 * lane; e < N * N; e += 256)
                            sad = __builtin_amdgcn_sad_u8(*reinterpret_cast<const uint32_t *>(&pr[e]), *reinterpret_cast<const uint32_t *>(&L.src[(st.y + (e >> lgN)) * 64 + st.x + (e & (N - 1))]), sad);
                    } else { /* chroma blocks are 4 .. 32 samples wide: rows of words */
                        const uint8_t *sc = &M.V.src_c[pl - 1][(st.y >> 1) * 32 + (st.x >> 1)];
                        for (int e = 4 * lane; e < n * n; e += 256)
                            sad = __builtin_amdgcn_sad_u8(*reinterpret_cast<const uint32_t *>(&pr[e]), *reinterpret_cast<const uint32_t *>(&sc[(e >> lgn) * 32 + (e & (n - 1))]), sad);
                    }
                } else if (luma) {
                    sad = md_intra_block((int)((cw0 >> 8) & 0xFF), N, lgN, M.ref, 1, lane, &L.src[st.y * 64 + st.x], 64, nullptr);
                } else { /* IntraPredictionOl's chroma pair: the chroma mode is always DM (Codec/EbIntraPrediction.c:5530) */
                    sad = md_intra_block((int)((cw0 >> 8) & 0xFF), N >> 1, lgN - 1, M.V.refc[pl - 1], 0, lane, &M.V.src_c[pl - 1][(st.y >> 1) * 32 + (st.x >> 1)], 32, nullptr);
                }
                sad = md_wave_sum(sad);
                MD_TR(23);
                MD_SUB(8);
                if (lane == 0) { /* every (candidate, plane, tile) has ONE owner: plain stores, nothing to initialise */
                    if (luma)
                        M.sadt[ci][ti] = sad;
                    else
                        M.V.sadc2[ci][pl - 1] = sad;
                }
                MD_TR(24);
                if (!draw)
                    break; /* (at most four tasks: this wave's one is done) */
            }
        }
        __syncthreads();
        MD_TR(25);
        MD_PROF(3);
        /* ================= C: fast costs (a lane per candidate), candidate buffers (a lane per buffer), PreModeDecision - in every wave ================= */
        uint32_t cst = 0xFFFFFFFFu; /* (a candidate the loop does not evaluate: the all-ones cost of an unused buffer) */
        if (in && evl) {
            uint32_t dist, distc = 0;
            if (heavy)
                dist = tiled64 ? M.sadt[lane][0] + M.sadt[lane][1] + M.sadt[lane][2] + M.sadt[lane][3] : M.sadt[lane][0];
            else
                dist = c.me_dist;
            /* md_inter_fast_cost_c / md_intra_fast_cost_pslice_c (md_logic.h) with the rate term of the first barrier's side */
            if (cfull) { /* the chroma pair's SAD with the noise-class rule (:2079-2094) */
                distc = (uint32_t)md_fast_chroma_noise_rule(&Lh, N, &c, (uint64_t)M.V.sadc2[lane][0] + M.V.sadc2[lane][1]);
                if (c.type == MD_INTER && c.merge_flag && Lh.cmplx_noise)
                    cst = ((dist + distc) << 8) + rterm; /* weightChromaDistortion == 0: a merge candidate's chroma SAD is added unweighted */
                else
                    cst = (dist << 8) + (uint32_t)md_weighted_chroma(distc, M.V.X.chroma_weight) + rterm;
            } else {
                cst = (dist << 8) + rterm;
            }
        } else {
            rate = 0; /* fastLumaRate of a candidate the loop does not evaluate (the reference never computes it) */
        }
        MD_TR(26);
        MD_SUB(9);
        /* md_fast_loop_buffers (md_logic.h; ProductPerformFastLoop's second loop, :1990-2179) with the buffers in lanes 0..7: the candidates arrive from the last to the
         * first, each goes into the buffer with the highest cost (an unused one first) = the FIRST buffer holding the maximum over [0, maxBuffers) */
        uint32_t bcost = 0xFFFFFFFFu;
        int bcand = -1, bpred = -1, evcount = 0;
        {
            int highest = 0;
            const int maxb = max_buffers < 2 ? 2 : max_buffers;
            const bool inb = lane < maxb;
            for (int idx = nc - 1; idx >= 0; idx--) {
                const uint32_t cv = md_rl(cst, idx);
                const int ev = __builtin_amdgcn_readlane(evl, idx);
                if (lane == highest) {
                    bcand = idx;
                    if (ev) {
                        bcost = cv;
                        if (!(ev & 2))
                            bpred = idx;
                    }
                }
                evcount += ev != 0;
                if (idx) { /* the first of lanes [0, maxb) holding their maximum: three DPP steps over the eight lanes, one ballot */
                    static_assert(MD_MAX_BUF == 8, "the buffers are the first eight lanes of a row");
                    const uint32_t v = inb ? bcost : 0u;
                    uint32_t m = v;
                    m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, false));  /* quad_perm [1,0,3,2] */
                    m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, false));  /* quad_perm [2,3,0,1] */
                    m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xF, 0xF, false)); /* row_half_mirror */
                    highest = __ffsll((long long)__ballot(inb && bcost == md_rl(m, 0))) - 1;
                }
            }
        }
        MD_TR(27);
        MD_SUB(10);
        /* PreModeDecision (md_pre_mode_decision, md_logic.h; Codec/EbModeDecision.c:300-383) on uniform values: the buffers' costs by v_readlane, the order as nibbles of a word */
        uint32_t best = 0; /* best[f] = (best >> 4 f) & 15 */
        int full_count, nfull;
        /* lane b: the type of buffer b's candidate.  (The shuffle runs in EVERY lane: a lane that sits the instruction out returns zero to whoever reads it - and the
         * candidates beyond the buffer count live in exactly the lanes that hold no buffer.) */
        const int btype_any = __shfl(ctype, bcand < 0 ? 0 : bcand), btype = bcand >= 0 ? btype_any : 0;
        {
            int bt = evcount < bufferTotal ? evcount : bufferTotal;
            const int same = evcount == bt, count = same ? bt : max_buffers;
            const int fullRecon = same ? (count < 1 ? 1 : count) : (count - 1 < 1 ? 1 : count - 1);
            if (count > 1) {
                int skipIdx = -1;
                if (!same) {
                    uint32_t hcost = md_rl(bcost, 0);
                    skipIdx = 0;
                    for (int i = 1; i < count; i++) {
                        const uint32_t v = md_rl(bcost, i);
                        if (v >= hcost)
                            hcost = v, skipIdx = i;
                    }
                }
                int k = 0;
                for (int i = 0; i < count; i++)
                    if (i != skipIdx)
                        best |= (uint32_t)i << (4 * k++);
            }
            const unsigned long long interm = __ballot(btype == MD_INTER) & 0xFFull, intram = __ballot(btype == MD_INTRA) & 0xFFull;
            for (int i = 0; i < fullRecon - 1; i++) /* inter candidates first */
                for (int j = i + 1; j < fullRecon; j++) {
                    const uint32_t bi = (best >> (4 * i)) & 15u, bj = (best >> (4 * j)) & 15u;
                    if (((intram >> bi) & 1ull) && ((interm >> bj) & 1ull))
                        best = (best & ~((15u << (4 * i)) | (15u << (4 * j)))) | (bj << (4 * i)) | (bi << (4 * j));
                }
            full_count = fullRecon;
            nfull = full_count < bt ? full_count : bt;
        }
        MD_TR(28);
        MD_PROF(4);
        /* ---- full loop: a wave per surviving candidate (PerformFullLoop, :4351) ---- */
        /* a 64x64 unit has four 32x32 transform units per candidate: a wave per (candidate, transform unit).  The candidates of a 64x64 unit are motion-compensated. */
        const bool split64 = N == 64 && nfull <= 4 && !any_intra;
        bool fresh64 = false;
        auto buf_cand = [&](int b) { return __builtin_amdgcn_readlane(bcand, b); };
        auto buf_pred = [&](int b) { return __builtin_amdgcn_readlane(bpred, b); };
        auto kept_pred = [&](int pci) { return (int)md_rl((uint32_t)slot, pci) >= 0 && __builtin_amdgcn_readlane(evl, pci) != 0; };
        if (split64) {
            for (int f = 0; f < nfull; f++) {
                const int b = (int)((best >> (4 * f)) & 15u), ci = buf_cand(b), pci = buf_pred(b) < 0 ? ci : buf_pred(b);
                if (!kept_pred(pci)) {
                    fresh64 = true;
                    if (wave == f)
                        md_predict_inter_plane(M.V.refs, (int)(md_rl(cw.w[2], pci) & 0xFF), md_rl(cw.w[4], pci), md_rl(cw.w[5], pci), x0, y0, N, 0, lane, M.V.mc[wave], M.V.wpred[f], 0, 1,
                                               &M.V.rw, &M.V.rwc);
                }
            }
            if (fresh64)
                __syncthreads();
            for (int f = 0; f < nfull; f++) {
                const int b = (int)((best >> (4 * f)) & 15u), ci = buf_cand(b), pci = buf_pred(b) < 0 ? ci : buf_pred(b);
                const uint8_t *pred = kept_pred(pci) ? M.V.cpred[(int)md_rl((uint32_t)slot, pci)] : M.V.wpred[f];
                const int tu = wave, off = ((tu & 1) << 5) + ((tu >> 1) << 5) * 64;
                const uint32_t c0 = md_rl(cw.w[0], ci);
                const MdFl o = md_full_loop_unit<32>(lane, &L.src[st.y * 64 + st.x] + off, 64, pred + off, 64, nullptr, M.tiles[wave], M.qbuf[wave], Ph.qp, Ph.slice_type, M.cost,
                                                     (int)(c0 & 0xFF), (int)((c0 >> 8) & 0xFF), 0, pf, M.rt);
                if (lane == 0)
                    M.fl[b][tu] = o;
                EP_WAVE_SYNC();
            }
        }
        for (int f = wave; f < nfull && !split64; f += 4) {
            MD_TR(30);
            const int b = (int)((best >> (4 * f)) & 15u), ci = buf_cand(b);
            const uint32_t c0 = md_rl(cw.w[0], ci);
            const int cdtype = (int)(c0 & 0xFF), cdmode = (int)((c0 >> 8) & 0xFF);
            /* the buffer's luma prediction: the candidate the fast loop predicted there, or - predictionIsReadyLuma == 0 - a fresh one */
            const bool fresh = ci == bestFirst && cdtype == MD_INTRA && __builtin_amdgcn_readlane(evl, ci) != 0;
            const int pci = (fresh || buf_pred(b) < 0) ? ci : buf_pred(b);
            const uint32_t p0 = md_rl(cw.w[0], pci);
            uint8_t *pred = M.V.wpred[wave];
            if ((p0 & 0xFF) == MD_INTER) {
                if (kept_pred(pci))
                    pred = M.V.cpred[(int)md_rl((uint32_t)slot, pci)]; /* the fast loop's prediction of this candidate is still there */
                else
                    md_predict_inter_plane(M.V.refs, (int)(md_rl(cw.w[2], pci) & 0xFF), md_rl(cw.w[4], pci), md_rl(cw.w[5], pci), x0, y0, N, 0, lane, M.V.mc[wave], pred, 0, 1, &M.V.rw, &M.V.rwc);
            } else {
                md_intra_block((int)((p0 >> 8) & 0xFF), N, lgN, M.ref, 1, lane, nullptr, 0, pred);
                EP_WAVE_SYNC();
            }
            MD_TR(31);
            MD_SUB(12);
            md_full_loop_cand(lane, N, &L.src[st.y * 64 + st.x], pred, N, nullptr, M.tiles[wave], M.qbuf[wave], P, M.cost, M.rt, cdtype, cdmode, pf, M.fl[b]);
            MD_TR(32);
            MD_SUB(13);
        }
        /* CHROMA_MODE_FULL (PerformFullLoop :4443-4560): the chroma pair of every survivor - ChromaPrediction (the candidate's OWN prediction: the fast loop's when it
         * evaluated the candidate, a fresh one otherwise), FullLoop_R + CuFullDistortionFastTuMode_R - as tasks (survivor, plane) on the waves the luma units left idle */
        if (cfull) {
            if (fresh64) /* every wave is done with the fresh luma predictions in the waves' scratch before a chroma block lands there */
                __syncthreads();
            const int Cn = N >> 1, lgc = lgN - 1, Tc = N == 64 ? 16 : Cn, ntu = N == 64 ? 4 : 1;
            /* the waves DRAW the chroma tasks as they get free: a luma unit with levels to price takes twice the time of one without, a wave without a luma unit starts at
             * once - no fixed assignment fits.  A task = (survivor, plane); the 8x8 chroma pair of a 16x16 unit's survivor is ONE task on the matrix cores (md_chroma_pair8). */
            const bool pair8 = N == 16 && !s_md_force_bfly;
            const int ntk = pair8 ? nfull : 2 * nfull;
            for (;;) {
                unsigned got = 0;
                if (lane == 0)
                    got = atomicAdd(&M.V.task_ctr2, 1u);
                const int tk = (int)md_rl(got, 0);
                if (tk >= ntk)
                    break;
                const int f = pair8 ? tk : tk >> 1, pl0 = pair8 ? 0 : tk & 1, pl1 = pair8 ? 2 : pl0 + 1, b = (int)((best >> (4 * f)) & 15u), ci = buf_cand(b);
                const uint32_t c0 = md_rl(cw.w[0], ci);
                const int cdtype = (int)(c0 & 0xFF), cdmode = (int)((c0 >> 8) & 0xFF);
                const uint8_t *pred; /* plane 0's block; plane 1's 1024 bytes behind */
                if (cdtype == MD_INTER && kept_pred(ci)) {
                    pred = M.V.cpred_c[(int)md_rl((uint32_t)slot, ci)][0];
                } else {
                    uint8_t *pw0 = M.V.wpred_c(wave, 0);
                    for (int pl = pl0; pl < pl1; pl++) {
                        uint8_t *pw = pw0 + pl * 1024;
                        if (cdtype == MD_INTER) {
                            md_predict_inter_plane(M.V.refs, (int)(md_rl(cw.w[2], ci) & 0xFF), md_rl(cw.w[4], ci), md_rl(cw.w[5], ci), x0, y0, N, 1 + pl, lane, M.V.mc[wave], pw, 0, 1, &M.V.rw,
                                                   &M.V.rwc);
                        } else {
                            md_intra_block(cdmode, Cn, lgc, M.V.refc[pl], 0, lane, nullptr, 0, pw);
                            EP_WAVE_SYNC();
                        }
                    }
                    pred = pw0;
                }
                if (pair8) {
                    md_chroma_pair8(lane, &M.V.src_c[0][(st.y >> 1) * 32 + (st.x >> 1)], pred, M.qbuf[wave], (int)P.chroma_qp, (int)P.slice_type, M.cost, cdtype, cdmode, pf, M.rt,
                                    &M.V.flc[b][0][0]);
                    continue;
                }
                const int pl = pl0;
                for (int tu = 0; tu < ntu; tu++) {
                    const int ox = ntu == 1 ? 0 : (tu & 1) << 4, oy = ntu == 1 ? 0 : (tu >> 1) << 4;
                    uint32_t nz;
                    unsigned long long d[2], bt;
                    md_chroma_tu(lane, Tc, &M.V.src_c[pl][((st.y >> 1) + oy) * 32 + (st.x >> 1) + ox], pred + pl * 1024 + oy * Cn + ox, Cn, M.tiles[wave], M.qbuf[wave], P, M.cost, M.rt,
                                 cdtype, cdmode, 1 + pl, pf, &nz, d, &bt);
                    if (lane == 0) {
                        MdFl o;
                        o.nz = nz, o.d0 = (uint32_t)d[0], o.d1 = (uint32_t)d[1], o.bits = (uint32_t)bt;
                        M.V.flc[b][pl][tu] = o;
                    }
                }
            }
        }
        MD_TR(33);
        __syncthreads();
        MD_TR(34);
        MD_PROF(5);
        /* ================= D (wave 0): TuCalcCostLuma + the full cost of every survivor (a lane each), ProductFullModeDecision, the depth decisions, the neighbour update ================= */
        if (wave == 0) {
            const bool have = lane < nfull;
            const int b = (int)((best >> (4 * (have ? lane : 0))) & 15u);
            const int ci = __shfl(bcand, b);
            uint32_t ycbf = 0;
            unsigned long long bits = 0, dist[2] = {0, 0}, full = 0;
            uint64_t mc = 0, sc = 0;
            const unsigned long long frate = __shfl(rate, have ? ci : 0); /* fastLumaRate of the survivor's candidate */
            MdCandWords sv; /* the survivor's candidate record */
            {
                const uint4 a = reinterpret_cast<const uint4 *>(&M.cand[have ? ci : 0])[0], bq = reinterpret_cast<const uint4 *>(&M.cand[have ? ci : 0])[1];
                sv.w[0] = a.x, sv.w[1] = a.y, sv.w[2] = a.z, sv.w[3] = a.w, sv.w[4] = bq.x, sv.w[5] = bq.y, sv.w[6] = bq.z, sv.w[7] = bq.w;
            }
            const int svtype = have ? sv.c.type : 0;
            if (have) {
                const MdCand &cs = sv.c;
                if (N == 64) {
                    for (int tu = 0; tu < 4; tu++)
                        md_tu_calc_cost(P, M.fl[b][tu], cs.type, 64, 32, tu + 1, &ycbf, &bits, dist);
                } else {
                    md_tu_calc_cost(P, M.fl[b][0], cs.type, N, N, 0, &ycbf, &bits, dist);
                }
                bits = md_pf_coeff_bits(pf, Ph.qp, bits); /* (CHROMA_MODE_BEST and CHROMA_MODE_FULL LCUs: the only ones md_lcu_supported admits) */
                if (cfull) { /* InterFullCost / MergeSkipFullCost / IntraFullCostPslice: the chroma loop's sums join the luma ones */
                    const int ntu = N == 64 ? 4 : 1;
                    uint32_t cbf[2] = {0, 0};
                    uint64_t cbits[2] = {0, 0}, cdist[2][2] = {{0, 0}, {0, 0}};
                    for (int pl = 0; pl < 2; pl++)
                        for (int tu = 0; tu < ntu; tu++) {
                            const MdFl o = M.V.flc[b][pl][tu];
                            cbf[pl] |= (uint32_t)(o.nz != 0) << (ntu == 1 ? 0 : tu + 1);
                            cbits[pl] += o.bits, cdist[pl][0] += o.d0, cdist[pl][1] += o.d1;
                        }
                    const uint64_t yd[2] = {dist[0], dist[1]};
                    if (cs.type == MD_INTER)
                        full = md_inter_full_cost(&P, M.V.X.chroma_weight, &cuv, &cs, N, ycbf, cbf, frate, yd, cdist, bits, cbits, &mc, &sc);
                    else
                        full = md_intra_full_cost_pslice(&P, M.V.X.chroma_weight, N, ycbf, cbf, frate, dist[0], cdist, bits, cbits);
                } else if (cs.type == MD_INTER) {
                    full = md_inter_full_luma_cost(&P, &cuv, &cs, N, ycbf, frate, (const uint64_t *)dist, bits, &mc, &sc);
                } else {
                    full = md_intra_full_luma_cost_pslice(&P, N, ycbf, frate, dist[0], bits);
                }
            }
            /* the reference walks the candidates in order: an intra candidate after an inter one whose root cbf is 0 is not costed at all (full-loop escape, :4450-4460) and
             * keeps whatever its buffer held (here: the all-ones cost of the buffer's initialisation) */
            uint32_t prevRootCbf = 1;
            unsigned long long bestFullCost = 0xFFFFFFFFull, kept = 0;
            for (int g = 0; g < nfull; g++) {
                const int ty = __builtin_amdgcn_readlane(svtype, g);
                const uint32_t yc = md_rl(ycbf, g);
                const unsigned long long cs_ = md_readlane64(full, g);
                if (ty == MD_INTRA && prevRootCbf == 0)
                    continue;
                kept |= 1ull << g;
                if (Ph.full_loop_escape && ty == MD_INTER && cs_ < bestFullCost)
                    prevRootCbf = yc, bestFullCost = cs_;
            }
            MD_TR(35);
            MD_SUB(14);
            /* ProductFullModeDecision (:1995): the lowest full cost among the first full_count buffers of the order (the first of equal ones) */
            const unsigned long long fcost = (have && ((kept >> lane) & 1ull)) ? full : ~0ull;
            int wf = 0;
            {
                unsigned long long lowestCost = ~0ull;
                for (int f = 0; f < full_count && f < nfull; f++) { /* (buffers beyond the survivors keep the all-ones cost: never lower) */
                    const unsigned long long v = md_readlane64(fcost, f);
                    if (v < lowestCost)
                        wf = f, lowestCost = v;
                }
            }
            /* the winner's lane hands its sums to lane 0 */
            const unsigned long long w_cost = md_readlane64(fcost, wf), w_mc = md_readlane64(mc, wf), w_sc = md_readlane64(sc, wf), w_bits = md_readlane64(bits, wf),
                                     w_d0 = md_readlane64(dist[0], wf), w_d1 = md_readlane64(dist[1], wf), w_rate = md_readlane64(frate, wf);
            const uint32_t w_ycbf = md_rl(ycbf, wf), w_c0 = md_rl(sv.w[0], wf), w_c2 = md_rl(sv.w[2], wf), w_mv0 = md_rl(sv.w[4], wf), w_mv1 = md_rl(sv.w[5], wf);
            const bool w_kept = ((kept >> wf) & 1ull) != 0;
            const int wtype = (int)(w_c0 & 0xFF), wdir = (int)(w_c2 & 0xFF);
            int last_v = leaf, upd_v = 0; /* (lane 0's; the wave reads them by v_readlane - a hand-over through LDS is two dependent round trips on the chain) */
            if (lane == 0) {
                MdCu &u = M.S.cu[leaf];
                /* (a winner the escape left uncosted keeps the buffer's initial values: zeros, as ProductResetModeDecision leaves them) */
                M.S.local[leaf].cost = w_cost, M.S.local[leaf].full_distortion = w_kept ? (uint32_t)w_d0 : 0u;
                u.pred_mode = (uint8_t)wtype, u.skip_flag = 0, u.intra_luma_mode = (uint8_t)(wtype == MD_INTRA ? ((w_c0 >> 8) & 0xFF) : 0x1F);
                const uint32_t yc = w_kept ? w_ycbf : 0u;
                u.ycbf = (uint8_t)(N == 64 ? (yc & 0x1E) : (yc & 1));
                { /* inter_dir | merge_flag | merge_index | pad, mv[0], mv[1]: three words of the record, three stores */
                    static_assert(offsetof(MdCu, inter_dir) == 8 && offsetof(MdCu, merge_flag) == 9 && offsetof(MdCu, merge_index) == 10 && offsetof(MdCu, mv) == 12 && sizeof(MdMv) == 4,
                                  "the unit's record, words 2..4");
                    const bool wi = wtype == MD_INTER;
                    uint32_t *q = reinterpret_cast<uint32_t *>(&u.inter_dir);
                    q[0] = (wi ? (uint32_t)wdir : 3u) | ((wi ? (w_c2 >> 8) & 0xFFu : 0u) << 8) | (((w_c2 >> 16) & 0xFFu) << 16);
                    q[1] = wi && wdir != MD_L1 ? w_mv0 : 0u, q[2] = wi && wdir != MD_L0 ? w_mv1 : 0u;
                }
                u.merge_cost = w_kept ? w_mc : 0, u.skip_cost = w_kept ? w_sc : 0;
                u.y_coeff_bits = w_kept ? w_bits : 0, u.y_dist[0] = w_kept ? w_d0 : 0, u.y_dist[1] = w_kept ? w_d1 : 0;
                u.fast_luma_rate = w_rate, u.ycbf_mask = yc;
                M.S.local[leaf].mdc_index = (uint8_t)cuIdx;
                int cur = leaf, curIdx = cuIdx, last;
                /* CheckHighCostPartition (md_check_high_cost_partition, md_logic.h) with everything it may read requested AT ONCE (the parent's record, the earlier siblings'
                 * costs, the counters of the depth decision, the next-unit step): one LDS round trip where the rule's early exits make a chain of five */
                const int off = md_depth_offset(st.depth), parent = st.parent;
                const MdLocal pl = M.S.local[parent];
                const uint64_t sib1 = M.S.local[leaf >= off ? leaf - off : 0].cost, sib2 = M.S.local[leaf >= 2 * off ? leaf - 2 * off : 0].cost;
                const int split_now = u.split, g8 = M.S.g8, g16 = M.S.g16, step = M.next_step[cuIdx];
                int exitParent = -1;
                if (Lh.is_complete && st.depth != 0 && st.ordinal < 4 && !split_now && pl.tested) {
                    const MdStats ps = md_stats(parent);
                    const int ctx = (pl.left_mode > MD_INTRA ? 0 : pl.left_depth > ps.depth) + (pl.top_mode > MD_INTRA ? 0 : pl.top_depth > ps.depth);
                    const uint64_t srate = ps.depth < 3 ? (ctx == 0 ? sfb0 : ctx == 1 ? sfb1 : sfb2) : 0;
                    const uint64_t parentCost = pl.cost + (((uint64_t)Ph.full_lambda * srate + (1u << 22)) >> 23);
                    const uint64_t children = w_cost + (st.ordinal >= 2 ? sib1 : 0) + (st.ordinal >= 3 ? sib2 : 0);
                    if (children > parentCost)
                        exitParent = parent;
                }
                if (exitParent >= 0) {
                    cur = exitParent, curIdx = pl.mdc_index;
                    M.S.cu[exitParent].split = 0;
                    last = md_inter_depth_decision(&P, &M.S, exitParent, lcu_x, lcu_y, 1, 0);
                } else if (st.ordinal < 4 || st.depth == 0) {
                    /* ProductPerformInterDepthDecision (md_inter_depth_decision) of a unit that is not the last of its four siblings: no depth is compared, the unit becomes a
                     * leaf when the leaf list or StopSplitCondition says so and the counters of finished blocks move */
                    if (split_now == 0 || ((w_kept ? (uint32_t)w_d0 : 0u) < (st.depth == 0 ? ss_thr0 : st.depth == 1 ? ss_thr1 : st.depth == 2 ? ss_thr2 : 0u))) {
                        u.split = 0;
                        if (st.depth == 1)
                            M.S.g16 = (uint8_t)(g16 + 1);
                        else if (st.depth == 2)
                            M.S.g8 = (uint8_t)(g8 + 1);
                    }
                    last = leaf;
                } else { /* open loop: no reconstruction to wait for, the inter-depth decision follows at once */
                    last = md_inter_depth_decision(&P, &M.S, leaf, lcu_x, lcu_y, 0, ((w_kept ? (uint32_t)w_d0 : 0u) < (st.depth == 0 ? ss_thr0 : st.depth == 1 ? ss_thr1 : st.depth == 2 ? ss_thr2 : 0u)));
                }
                last_v = last, upd_v = M.S.cu[last].split == 0;
                /* the next unit (CalculateNextCuIndex :1261): the loop stands on `cur` - the tested unit, or the parent a partition exit fell back to */
                int nextIdx = curIdx;
                if (M.S.cu[cur].split || lh < 64)
                    nextIdx++;
                else
                    nextIdx += cur == leaf ? step : (int)M.next_step[curIdx];
                M.cu_idx = nextIdx;
                M.done = nextIdx >= Lh.leaf_count;
#ifdef MD_TRACE
                g_md_trace_on = D.trace && lcu == D.trace_lcu && nextIdx >= D.trace_unit && nextIdx < D.trace_unit + 2;
#endif
            }
            MD_TR(36);
            EP_WAVE_SYNC();
            MD_PROF(6);
            /* ModeDecisionUpdateNeighborArrays of the unit the decision ended on (at most 256 cells: this wave's lanes) */
            if (md_rl((uint32_t)upd_v, 0)) {
                const int last = (int)md_rl((uint32_t)last_v, 0);
                uint32_t w, m0, m1, md;
                MdStats ls = st;
                if (last == leaf) { /* the unit just decided (most updates): its record is in this wave's registers */
                    const bool wi = wtype == MD_INTER;
                    w = (uint32_t)wtype | ((wtype == MD_INTRA ? (w_c0 >> 8) & 0xFFu : 0x1Fu) << 8) | ((uint32_t)st.depth << 16);
                    m0 = wi && wdir != MD_L1 ? w_mv0 : 0u, m1 = wi && wdir != MD_L0 ? w_mv1 : 0u, md = wi ? (uint32_t)wdir : 3u;
                } else {
                    ls = md_stats(last);
                    const MdCu u = M.S.cu[last];
                    w = (uint32_t)u.pred_mode | ((uint32_t)u.intra_luma_mode << 8) | ((uint32_t)ls.depth << 16) | ((uint32_t)u.skip_flag << 24);
                    m0 = md_pack_mv(u.mv[0]), m1 = md_pack_mv(u.mv[1]), md = u.inter_dir;
                }
                const int cells = ls.size >> 2, lgc4 = ls.lg - 2;
                for (int e = lane; e < cells * cells; e += 64)
                    L.info[((ls.y >> 2) + (e >> lgc4) + 1) * 36 + (ls.x >> 2) + (e & (cells - 1)) + 1] = w;
                const int c8 = ls.size >> 3, lgc8 = ls.lg - 3;
                static_assert(sizeof(MdMvUnit) == 12, "three words: mv[0], mv[1], dir | avail << 8");
                for (int e = lane; e < c8 * c8; e += 64) {
                    uint32_t *q = reinterpret_cast<uint32_t *>(&M.V.mvu[((ls.y >> 3) + (e >> lgc8) + 1) * 18 + (ls.x >> 3) + (e & (c8 - 1)) + 1]);
                    q[0] = m0, q[1] = m1, q[2] = md;
                }
            }
            MD_TR(38);
        }
        __syncthreads();
        MD_TR(39);
        MD_PROF(8);
        cuIdx = M.cu_idx; /* (the next unit, or the end of the leaf list: one LDS round trip for both) */
        if (cuIdx >= (int)Lh.leaf_count)
            break;
    }
}

/* ModeDecisionLcu of one LCU: on return M.S holds the decisions, the picture's maps the LCU's final neighbour state */
/* what the LCU's mode decision reads that no other LCU of the picture writes (its records, its source): into LDS BEFORE the workgroup waits for the LCU's neighbours */
template <bool INTER>
__device__ __forceinline__ void md_lcu_inputs(const MdPictureDev &D, const SvtAmdMdPicture &P, int lcu, int lcu_x, int lcu_y, MdShared<INTER> &M)
{
    const int t = threadIdx.x;
    auto &L = M.L;
    const int lw = min(64, (int)P.width - lcu_x), lh = min(64, (int)P.height - lcu_y);
    for (int i = t; i < (int)sizeof(SvtAmdMdLcu); i += 256)
        ((uint8_t *)&M.lcu)[i] = ((const uint8_t *)&D.lcus[lcu])[i];
    static_assert(sizeof(SvtAmdCabacCost) % 4 == 0 && sizeof(SvtAmdMdPicture) % 4 == 0, "record sizes");
    for (int i = t; i < (int)(sizeof(SvtAmdMdPicture) / 4); i += 256)
        ((uint32_t *)&M.pic)[i] = ((const uint32_t *)D.P)[i];
    if (D.cost)
        for (int i = t; i < (int)(sizeof(SvtAmdCabacCost) / 4); i += 256)
            ((uint32_t *)&M.cost)[i] = ((const uint32_t *)D.cost)[i];
    static_assert(sizeof(RateTables) % 4 == 0, "record sizes");
    for (int i = t; i < (int)(sizeof(RateTables) / 4); i += 256)
        ((uint32_t *)&M.rt)[i] = ((const uint32_t *)&c_rt)[i];
    static_assert(sizeof(SvtAmdOisLcuResult) % 4 == 0 && sizeof(SvtAmdMeLcuResult) % 4 == 0 && sizeof(SvtAmdMeCuResult) % 4 == 0 && sizeof(SvtAmdTmvpLcu) % 8 == 0, "record sizes");
    for (int i = t; i < (int)(sizeof(SvtAmdOisLcuResult) / 4); i += 256)
        ((uint32_t *)&M.ois)[i] = ((const uint32_t *)&D.ois[lcu])[i];
    if constexpr (INTER) {
        for (int i = t; i < (int)(sizeof(M.V.me) / 4); i += 256)
            ((uint32_t *)M.V.me)[i] = ((const uint32_t *)D.me[lcu].pu)[i];
        static_assert(sizeof(EpRefPlanes) % 4 == 0, "record sizes");
        for (int i = t; i < (int)(2 * sizeof(EpRefPlanes) / 4); i += 256)
            ((uint32_t *)M.V.refs)[i] = ((const uint32_t *)D.mref)[i];
        static_assert(sizeof(SvtAmdMdInter) % 4 == 0, "record sizes");
        for (int i = t; i < (int)(sizeof(SvtAmdMdInter) / 4); i += 256)
            ((uint32_t *)&M.V.X)[i] = ((const uint32_t *)D.X)[i];
        if (D.X->tmvp_enable)
            for (int i = t; i < (int)(2 * sizeof(SvtAmdTmvpLcu) / 4); i += 256)
                ((uint32_t *)M.V.tmvp)[i] = ((const uint32_t *)&D.tmvp[lcu])[i];
    }
    for (int i = t; i < 64 * 64 / 4; i += 256) {
        const int y = i >> 4, x = (i & 15) * 4;
        uint32_t v = 0;
        if (x < lw && y < lh)
            v = *(const uint32_t *)(D.src[0] + (size_t)(lcu_y + y) * D.src_pitch[0] + lcu_x + x);
        *(uint32_t *)&L.src[y * 64 + x] = v;
    }
    if constexpr (INTER) { /* the chroma source: CHROMA_MODE_FULL candidates, and the merge / skip decisions behind the mode decision */
        for (int i = t; i < 2 * 32 * 32 / 4; i += 256) {
            const int p = i >> 8, e = i & 255, y = e >> 3, x = (e & 7) * 4;
            uint32_t v = 0;
            if (x < lw / 2 && y < lh / 2)
                v = *(const uint32_t *)(D.src[1 + p] + (size_t)(lcu_y / 2 + y) * D.src_pitch[1] + lcu_x / 2 + x);
            *(uint32_t *)&M.V.src_c[p][y * 32 + x] = v;
        }
    }
    __syncthreads();
    if (t == 0) {
        md_construct_cu_array(&M.S, &M.lcu);
        M.cu_idx = 0, M.done = 0;
    }
    if (t >= 64 && t < 64 + (int)M.lcu.leaf_count) /* (before the wait for the neighbours: off the chain) */
        M.next_step[t - 64] = (uint8_t)md_next_cu_step(&M.lcu, t - 64, md_stats(M.lcu.leaf_index[t - 64]).depth);
    if constexpr (INTER) {
        static_assert(sizeof(MdStats) == 8, "two words");
        if (t >= 128 && t < 128 + (int)M.lcu.leaf_count) {
            const int lf = M.lcu.leaf_index[t - 128];
            const MdStats s_ = md_stats(lf);
            uint32_t w_[2];
            __builtin_memcpy(w_, &s_, 8);
            M.V.unit_tab[t - 128] = make_uint4(w_[0], w_[1], (uint32_t)lf, 0u);
        }
    }
    if constexpr (INTER) {
        /* the spatial neighbours of every unit of the leaf list with the availability GenerateL0L1AmvpMergeLists derives from positions (EbAdaptiveMotionVectorPrediction.c:2256-2340:
         * scan order, array bounds, tile edges); whether the neighbour is an inter unit is the one thing left to the unit's own time */
        const bool tl = M.lcu.tile_left != 0, tt = M.lcu.tile_top != 0, tr = M.lcu.tile_right != 0;
        for (int i = t; i < 5 * (int)M.lcu.leaf_count; i += 256) {
            const int ci = i / 5, k = i - 5 * ci;
            const MdStats st = md_stats(M.lcu.leaf_index[ci]);
            const int N = st.size;
            const bool left = tl && st.x == 0, top = tt && st.y == 0, right = tr && ((st.x + N) & 63) == 0;
            const int px = k == 2 ? st.x + N : k == 3 ? st.x + N - 1 : st.x - 1, py = k == 0 ? st.y + N : k == 1 ? st.y + N - 1 : st.y - 1;
            bool ok = k == 0 ? md_bottom_left_ok(&st) && !left : k == 1 ? !left : k == 2 ? md_top_right_ok(&st) && !top && !right : k == 3 ? !top : !left && !top;
            const int cx = px >> 2, cy = py >> 2;
            if (cy >= 16 || cx >= 32 || (cy >= 0 && cx >= 16)) /* (MdLocal8T::info_at: never written before this LCU's units) */
                ok = false;
            const int ii = ok ? (cy + 1) * 36 + cx + 1 : 0, mi = ok ? ((py >> 3) + 1) * 18 + (px >> 3) + 1 : 0;
            M.V.nbtab[ci][k] = (uint32_t)ii | ((uint32_t)mi << 10) | ((uint32_t)ok << 18);
        }
    }
    if constexpr (INTER) { /* still before the wait for the LCU's neighbours: the reference samples its candidates will most likely read */
        const SvtAmdMeCuResult me0 = M.V.me[0];
        int16_t cmv[2][2];
        cmv[0][0] = me0.x_mv_l0, cmv[0][1] = me0.y_mv_l0, cmv[1][0] = me0.x_mv_l1, cmv[1][1] = me0.y_mv_l1;
        const bool use[2] = {true, P.slice_type == 0};
        ep_ref_windows_fill(D.mref, lcu_x, lcu_y, use, cmv, M.V.rw, t);
        const bool usec[2] = {M.lcu.chroma_encode_mode == 1, M.lcu.chroma_encode_mode == 1 && P.slice_type == 0};
        ep_ref_windows_fill_chroma(D.mref, lcu_x, lcu_y, usec, cmv, M.V.rwc, t);
    }
}

/* ModeDecisionLcu of one LCU (its inputs are in LDS: md_lcu_inputs): on return M.S holds the decisions, the picture's maps the LCU's final neighbour state */
template <bool INTER, bool PROF>
__device__ __forceinline__ void md_lcu(const MdPictureDev &D, const SvtAmdMdPicture &Pg, int lcu, int lcu_x, int lcu_y, MdShared<INTER> &M)
{
    /* the picture's controls as the unit loop reads them: the copies md_lcu_inputs left beside the LCU in LDS, not the records in HBM (a unit's scalar stages read dozens
     * of these fields one after the other - each a round trip of its own from global memory) */
    const SvtAmdMdPicture &P = M.pic;
    (void)Pg;
    const bool prof_on = PROF && __builtin_amdgcn_readfirstlane((int)(D.prof != nullptr)) != 0;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    auto &L = M.L;
    const int W = (int)P.width, H = (int)P.height;
    const int lw = min(64, W - lcu_x), lh = min(64, H - lcu_y);
    const bool islice = P.slice_type == 2, open_loop = P.intra_md_open_loop != 0;
    /* ---- the LCU's surroundings (what its neighbours left in the picture's maps) into LDS ---- */
    for (int i = t; i < 17 * 36; i += 256) {
        const int cy = i / 36 - 1, cx = i - (cy + 1) * 36 - 1;
        uint32_t v = 0xFFFFFFFFu;
        if (cy < 0 || cx < 0) {
            const int px = lcu_x + 4 * cx, py = lcu_y + 4 * cy;
            v = (px < 0 || py < 0 || px >= W || py >= H) ? 0xFFFFFFFEu : D.md_info[(size_t)(py >> 2) * D.info_pitch + (px >> 2)];
        }
        L.info[i] = v;
    }
    if (!INTER) {
        for (int i = t; i < 130 + 64; i += 256) { /* ring samples: row -1 (x = -1 .. 128), column -1 */
            const int x = i < 130 ? i - 1 : -1, y = i < 130 ? -1 : i - 130;
            const int gx = lcu_x + x, gy = lcu_y + y;
            if (x < 128 && gx >= 0 && gy >= 0 && gx < W && gy < H)
                *L.at(x, y) = D.md_rec[(size_t)gy * D.md_pitch + gx];
        }
    }
    if constexpr (INTER) {
        for (int i = t; i < 9 * 18; i += 256) { /* the ring of motion-vector units: row -1 and column -1 */
            const int cy = i / 18 - 1, cx = i - (cy + 1) * 18 - 1;
            MdMvUnit u;
            u.mv[0].x = u.mv[0].y = u.mv[1].x = u.mv[1].y = 0, u.dir = 0, u.avail = 0, u.pad[0] = u.pad[1] = 0;
            const int px = lcu_x + 8 * cx, py = lcu_y + 8 * cy;
            if ((cy < 0 || cx < 0) && px >= 0 && py >= 0 && px < W && py < H) {
                const uint4 w = D.md_mv[(size_t)(py >> 3) * D.mv_pitch + (px >> 3)];
                u.mv[0].x = (int16_t)(w.x & 0xFFFF), u.mv[0].y = (int16_t)(w.x >> 16), u.mv[1].x = (int16_t)(w.y & 0xFFFF), u.mv[1].y = (int16_t)(w.y >> 16);
                u.dir = (uint8_t)w.z;
            }
            M.V.mvu[i] = u;
        }
    }
    __syncthreads();
    if constexpr (INTER) {
        /* The staged reference samples were centred on the 64x64 unit's motion-estimation vector before the wait.  The candidates the chain really predicts are mostly merge
         * candidates = the neighbours' vectors; where those lie elsewhere (periodic or low-texture content: the open-loop search and the decisions disagree) the window of
         * that list is staged again around the vector of the LCU's left / top neighbour - once per LCU instead of a round trip to HBM per candidate of every unit. */
        MdMvUnit nbu = M.V.mvu[1 * 18 + 0]; /* (x = -1, y = 0) */
        int have = (L.info_at(-1, 0) & 0xFF) == MD_INTER && !M.lcu.tile_left;
        if (!have) {
            nbu = M.V.mvu[0 * 18 + 1]; /* (x = 0, y = -1) */
            have = (L.info_at(0, -1) & 0xFF) == MD_INTER && !M.lcu.tile_top;
        }
        bool use[2] = {false, false};
        int16_t cmv[2][2] = {{0, 0}, {0, 0}};
        if (have) {
            const SvtAmdMeCuResult me0 = M.V.me[0];
            const int mex[2] = {me0.x_mv_l0, me0.x_mv_l1}, mey[2] = {me0.y_mv_l0, me0.y_mv_l1};
            for (int l = 0; l < (P.slice_type == 0 ? 2 : 1); l++)
                if ((nbu.dir == MD_BI || nbu.dir == l) && (abs(nbu.mv[l].x - mex[l]) > MD_RESTAGE_THR || abs(nbu.mv[l].y - mey[l]) > MD_RESTAGE_THR))
                    use[l] = true, cmv[l][0] = nbu.mv[l].x, cmv[l][1] = nbu.mv[l].y;
        }
        if (use[0] || use[1]) { /* (uniform: every thread read the same LDS words) */
            const int x0k[2] = {M.V.rw.x0[0], M.V.rw.x0[1]}, y0k[2] = {M.V.rw.y0[0], M.V.rw.y0[1]}, vk[2] = {M.V.rw.valid[0], M.V.rw.valid[1]};
            __syncthreads();
            ep_ref_windows_fill(D.mref, lcu_x, lcu_y, use, cmv, M.V.rw, t);
            if (t == 0)
                for (int l = 0; l < 2; l++)
                    if (!use[l]) /* the list that keeps its window (the fill marks an unused list invalid) */
                        M.V.rw.x0[l] = x0k[l], M.V.rw.y0[l] = y0k[l], M.V.rw.valid[l] = vk[l];
            __syncthreads();
        }
    }
    MD_PROF(0);
    const SvtAmdOisLcuResult *ois = &M.ois;
    const int pf = md_pf_mode(&P);
    constexpr bool NEW_INTER_LOOP = INTER; /* P / B pictures: md_units_inter (round 6); the loop below is the I pictures' */
    if constexpr (NEW_INTER_LOOP)
        if (M.lcu.chroma_encode_mode == 1)
            md_units_inter<PROF, true>(D, lcu, lcu_x, lcu_y, M);
        else
            md_units_inter<PROF, false>(D, lcu, lcu_x, lcu_y, M);
    else
    for (;;) {
        MD_TR(10);
        /* ---- lane 0: the unit, its contexts and its candidates ---- */
        if (t == 0) {
            const int cuIdx = M.cu_idx, leaf = M.lcu.leaf_index[cuIdx];
            const MdStats st = md_stats(leaf);
            M.leaf = leaf;
            if (prof_on)
                M.prof_depth = st.depth;
            M.S.local[leaf].tested = 1;
            M.S.cu[leaf].split = (uint8_t)((islice && st.depth == 0) ? 1 : M.lcu.leaf_split[cuIdx]);
            uint32_t l = L.info_at(st.x - 1, st.y), tp = L.info_at(st.x, st.y - 1);
            if ((M.lcu.tile_left && st.x == 0) || (l & 0xFF) == 0xFE)
                l = 0xFFFFFFFFu;
            if ((M.lcu.tile_top && st.y == 0) || (tp & 0xFF) == 0xFE)
                tp = 0xFFFFFFFFu;
            MdNeighbors Nb;
            Nb.left_mode = (uint8_t)l, Nb.left_intra = (uint8_t)(l >> 8), Nb.left_depth = (uint8_t)(l >> 16), Nb.left_skip = (uint8_t)(l >> 24);
            Nb.top_mode = (uint8_t)tp, Nb.top_intra = (uint8_t)(tp >> 8), Nb.top_depth = (uint8_t)(tp >> 16), Nb.top_skip = (uint8_t)(tp >> 24);
            md_context_generation(&M.S, leaf, st.y, &Nb);
            M.S.cu[leaf].split = (uint8_t)md_skip_small_cu(&P, &M.lcu, &M.S, leaf, st.depth);
            int ncand = 0;
            if (st.depth != 0 && (islice || st.depth == 3 || !M.lcu.restrict_intra_global_motion))
                if (!(P.limit_intra && st.x == 0 && st.y == 0))
                    ncand = md_intra_candidates(&P, &M.lcu, ois, leaf, &st, M.cand);
            M.ncand = ncand; /* the intra candidates so far (P / B pictures: the lists below are made by three waves) */
        }
        if (wave == 0)
            MD_TR(11);
        MD_SUB(0);
        if constexpr (INTER) {
            /* GenerateL0L1AmvpMergeLists: the AMVP candidates of list 0, of list 1 and the merge candidates share their inputs and nothing else - and none of them needs
             * what lane 0 of the first wave derives meanwhile (contexts, intra candidates).  Waves 1 and 2 therefore start at the unit's first moment: five lanes of each
             * fetch the spatial neighbours with the availability GenerateL0L1AmvpMergeLists derives (EbAdaptiveMotionVectorPrediction.c:2256-2340) into the wave's OWN copy
             * (no workgroup barrier between the fetch and the list), lane 0 then makes the wave's lists - four chains side by side instead of a barrier after the first. */
            if (wave >= 1) {
                const MdStats st = md_stats(M.lcu.leaf_index[M.cu_idx]);
                if (wave < 3) {
                    /* the five spatial neighbours (A0, A1, B0, B1, B2): a lane each, then every lane of the wave holds all five IN REGISTERS (v_readlane) - the list code
                     * below runs on registers in every lane alike, with no trip through LDS between the fetch and the lists */
                    uint32_t w0 = 0, w1 = 0, w2 = 0;
                    if (lane < 5) {
                        const int N = st.size, k = lane;
                        const bool left = M.lcu.tile_left && st.x == 0, top = M.lcu.tile_top && st.y == 0, right = M.lcu.tile_right && ((st.x + N) & 63) == 0;
                        const int px = k == 2 ? st.x + N : k == 3 ? st.x + N - 1 : st.x - 1, py = k == 0 ? st.y + N : k == 1 ? st.y + N - 1 : st.y - 1;
                        const bool ok = k == 0 ? md_bottom_left_ok(&st) && !left : k == 1 ? !left : k == 2 ? md_top_right_ok(&st) && !top && !right : k == 3 ? !top : !left && !top;
                        if (ok && (L.info_at(px, py) & 0xFF) == MD_INTER) {
                            const uint32_t *q = reinterpret_cast<const uint32_t *>(M.V.mv_at(px, py));
                            w0 = q[0], w1 = q[1], w2 = (q[2] & 0xFFu) | 0x100u; /* mv[0], mv[1], dir | avail << 8 */
                        }
                    }
                    MdMvUnit nbr[5];
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)w0, k), b = (uint32_t)__builtin_amdgcn_readlane((int)w1, k),
                                       c = (uint32_t)__builtin_amdgcn_readlane((int)w2, k);
                        nbr[k].mv[0].x = (int16_t)(a & 0xFFFF), nbr[k].mv[0].y = (int16_t)(a >> 16), nbr[k].mv[1].x = (int16_t)(b & 0xFFFF), nbr[k].mv[1].y = (int16_t)(b >> 16);
                        nbr[k].dir = (uint8_t)(c & 0xFF), nbr[k].avail = (uint8_t)((c >> 8) & 1), nbr[k].pad[0] = nbr[k].pad[1] = 0;
                    }
                    MD_TR(12);
                    MdInterLists T;
                    T.amvp_count[0] = T.amvp_count[1] = 0, T.merge_count = 0, T.pad = 0;
                    const int ox = lcu_x + st.x, oy = lcu_y + st.y, totalMerge = md_nmm(&P, st.size);
                    /* wave 1: both AMVP lists, wave 2: the merge candidates (the longest of the three) */
                    md_amvp_merge_lists_parts(&P, &M.V.X, nbr, M.V.X.tmvp_enable ? M.V.tmvp : nullptr, ox, oy, st.size, totalMerge, &T, wave == 1 ? 3 : 4);
                    MD_TR(13);
                    bool keep = false;
                    MdCand c;
                    c.type = MD_INTER, c.intra_mode = 0, c.mpm = 0, c.dist_ready = 0, c.me_dist = 0, c.dir = 0, c.merge_flag = 0, c.merge_index = 0;
                    c.mvp_idx[0] = c.mvp_idx[1] = 0, c.pad[0] = c.pad[1] = c.pad[2] = 0;
                    c.mv[0].x = c.mv[0].y = c.mv[1].x = c.mv[1].y = 0, c.mvp[0].x = c.mvp[0].y = c.mvp[1].x = c.mvp[1].y = 0;
                    if (wave == 1) { /* Me2Nx2NCandidatesInjection: a lane per motion-estimation candidate (md_inter_candidates, md_logic.h) */
                        const SvtAmdMeCuResult *me = &M.V.me[md_raster_index(&st)];
                        if (lane < 3 && lane < me->total_me_candidate_index) {
                            const int dir = me->direction[lane];
                            if (!(dir == MD_BI && P.depth_mode == 0 && M.lcu.lcu_md_mode == 10)) {
                                keep = true;
                                c.dist_ready = 1, c.me_dist = me->distortion[lane], c.dir = (uint8_t)dir;
                                c.mv[0].x = me->x_mv_l0, c.mv[0].y = me->y_mv_l0, c.mv[1].x = me->x_mv_l1, c.mv[1].y = me->y_mv_l1;
                                md_choose_mvp(&P, (uint32_t)ox, (uint32_t)oy, &T, &c);
                            }
                        }
                    } else { /* ProductMergeSkip2Nx2NCandidatesInjection: a lane per merge candidate, duplicates of earlier ones dropped */
                        const int k = lane;
                        if (k < 5 && k < totalMerge && k < T.merge_count) {
                            const MdMergeCand mc = k == 0 ? T.merge[0] : k == 1 ? T.merge[1] : k == 2 ? T.merge[2] : k == 3 ? T.merge[3] : T.merge[4];
                            bool dup = false;
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const MdMergeCand d = T.merge[j];
                                const bool f0 = mc.mv[0].x == d.mv[0].x && mc.mv[0].y == d.mv[0].y;
                                const bool f1 = mc.dir != MD_L0 && mc.mv[1].x == d.mv[1].x && mc.mv[1].y == d.mv[1].y;
                                const bool same = mc.dir == MD_L0 ? f0 : (mc.dir == MD_L1 ? f1 : (f0 && f1));
                                dup = dup || (j < k && mc.dir == d.dir && same);
                            }
                            if (!dup) {
                                keep = true;
                                c.dir = mc.dir, c.merge_flag = 1, c.merge_index = (uint8_t)k, c.mv[0] = mc.mv[0], c.mv[1] = mc.mv[1];
                            }
                        }
                    }
                    const unsigned long long km = __ballot(keep);
                    if (keep)
                        (wave == 1 ? M.V.me_c : M.V.mg_c)[__popcll(km & ((1ull << lane) - 1ull))] = c;
                    if (lane == 0)
                        (wave == 1 ? M.V.n_me : M.V.n_mg) = __popcll(km);
                    MD_TR(16);
                }
                /* ... and the fourth wave the unit's intra reference (only units below 64x64 have intra candidates) - source samples from HBM in the open-loop decision, a
                 * round trip under the other waves' chains */
                if (wave == 3 && st.depth != 0) {
                    if (open_loop)
                        md_build_refs_ol(D, M, st, lcu_x + st.x, lcu_y + st.y, W, H, lane);
                    else
                        md_build_refs(M, st, lane);
                    if (M.lcu.chroma_encode_mode == 1 && open_loop)
                        md_build_refs_ol_chroma(D, M.V.refc, st.size, lcu_x + st.x, lcu_y + st.y, W, H, lane);
                }
                if (wave == 3)
                    MD_TR(14);
            }
            MD_SUB(2);
            __syncthreads();
            MD_TR(15);
            MD_SUB(3);
            /* wave 0 appends the two waves' candidates to the intra candidates: motion-estimation candidates first, merge candidates behind them (md_inter_candidates' order) */
            if (wave == 0) {
                const int n0 = M.ncand, nme = M.V.n_me, nmg = M.V.n_mg;
                if (lane < nme)
                    M.cand[n0 + lane] = M.V.me_c[lane];
                else if (lane < nme + nmg)
                    M.cand[n0 + lane] = M.V.mg_c[lane - nme];
                if (lane == 0)
                    M.ncand = n0 + nme + nmg;
                EP_WAVE_SYNC();
            }
        }
        MD_SUB(4);
        if (t == 0) {
            const int leaf = M.leaf;
            const MdStats st = md_stats(leaf);
            int ncand = M.ncand;
            uint32_t mpm[3] = {0, 0, 0};
            if (P.mpm_search && !M.lcu.restrict_intra_global_motion)
                md_mpm_modes(M.S.cu[leaf].left_intra_mode, M.S.cu[leaf].top_intra_mode, mpm);
            int bufferTotal = md_nfl(&P, &M.lcu, st.size);
            ncand = md_mpm_injection(&P, &M.lcu, &st, M.cand, ncand, &bufferTotal, mpm);
            bufferTotal = ncand < bufferTotal ? ncand : bufferTotal;
            const int width = st.depth == 0 ? 5 : 8;
            M.ncand = ncand, M.buffer_total = bufferTotal, M.max_buffers = bufferTotal + 1 < width ? bufferTotal + 1 : width;
        }
        if (wave == 0)
            MD_TR(17);
        MD_SUB(5);
        if (wave == 0) { /* a lane per candidate (MD_MAX_CAND <= 64): what the candidate list implies for the loops below */
            EP_WAVE_SYNC();
            const int nc = M.ncand, lf = M.leaf;
            const MdStats s1 = md_stats(lf);
            const bool in = lane < nc;
            MdCand c = M.cand[in ? lane : 0];
            M.any_intra = __ballot(in && c.type == MD_INTRA) != 0;
            /* the first fast loop (EbProductCodingLoop.c:1948-1988): the best of the candidates whose distortion the open-loop stages left; the reference
             * walks from the last candidate down with <=: the LOWEST index among equal costs */
            int bestFirst = -1;
            if (!P.single_fast_loop) {
                const bool ready = in && c.dist_ready;
                unsigned long long cost = ~0ull;
                if (ready) {
                    uint64_t r;
                    cost = c.type == MD_INTER ? md_inter_fast_cost(&P, &s1, &M.S.cu[lf], &c, c.me_dist, &r)
                           : islice          ? md_intra_fast_cost_islice(&P, &s1, &M.S.cu[lf], c.intra_mode, c.me_dist, &r)
                                             : md_intra_fast_cost_pslice(&P, &s1, &M.S.cu[lf], c.intra_mode, c.me_dist, &r);
                }
                /* few candidates are ready: a scalar walk over them instead of a 64-lane reduction */
                unsigned long long m = ~0ull, rm = __ballot(ready);
                while (rm) {
                    const int l = __ffsll((long long)rm) - 1;
                    rm &= rm - 1;
                    const unsigned long long v = md_readlane64(cost, l);
                    if (bestFirst < 0 || v < m)
                        m = v, bestFirst = l;
                }
            }
            uint8_t e = (uint8_t)(in && (!c.dist_ready || lane == bestFirst || P.single_fast_loop));
            if (e && lane == bestFirst && c.type == MD_INTRA && open_loop)
                e = 3; /* the open-loop distortion stands, no luma prediction (:1660, :2042) */
            if (in)
                M.evaluated[lane] = e;
            {   /* what the fast loop really has to do: predict + measure the evaluated candidates that are neither most-probable-mode placeholders nor the open-loop
                 * intra candidate whose distortion stands (:2042).  Most P / B candidates are not among them (the motion-estimation candidates bring their distortion:
                 * only the best of them is evaluated), so the list is packed - a wave per LIST entry keeps all four waves on real work */
                const bool heavy = in && e && !c.mpm && !(lane == bestFirst && c.type == MD_INTRA);
                const unsigned long long hm = __ballot(heavy);
                if (heavy)
                    M.heavy[__popcll(hm & ((1ull << lane) - 1ull))] = (uint8_t)lane;
                if (in) /* the heavy candidates' distortion is summed up by the fast loop (several waves may add to it) */
                    M.sad[lane] = (!heavy && e && !c.mpm) ? c.me_dist : 0u;
                if (lane == 0)
                    M.nheavy = __popcll(hm);
                if constexpr (INTER) { /* CHROMA_MODE_FULL: the chroma pair of EVERY evaluated candidate is predicted and measured - the open-loop intra candidate that won
                                        * the first loop included (only its luma distortion stands, :1651-1654) */
                    const bool hc = in && e && !c.mpm && M.lcu.chroma_encode_mode == 1;
                    const unsigned long long cm = __ballot(hc);
                    if (hc)
                        M.V.heavyc[__popcll(cm & ((1ull << lane) - 1ull))] = (uint8_t)lane;
                    if (in)
                        M.V.sadc[lane] = 0u;
                    if (lane == 0)
                        M.V.nheavyc = __popcll(cm);
                }
            }
            if constexpr (INTER) { /* the first eight inter candidates the loop evaluates keep their prediction for the full loop */
                const bool q = e && c.type == MD_INTER;
                const unsigned long long qm = __ballot(q);
                const int rank = __popcll(qm & ((1ull << lane) - 1ull));
                if (in)
                    M.V.slot[lane] = (int8_t)((q && rank < MD_PRED_SLOTS) ? rank : -1);
            }
            if (lane == 0)
                M.best_first = bestFirst;
            MD_TR(18);
        }
        MD_SUB(6);
        __syncthreads();
        MD_TR(19);
        MD_PROF(1);
        const int leaf = M.leaf, ncand = M.ncand;
        const MdStats st = md_stats(leaf);
        const int N = st.size, lgN = st.lg, x0 = lcu_x + st.x, y0 = lcu_y + st.y;
        /* ---- wave 0: the unit's intra reference (P / B pictures: made beside the motion-vector lists above) ---- */
        if constexpr (!INTER) {
            if (wave == 0 && M.any_intra) {
                if (open_loop)
                    md_build_refs_ol(D, M, st, x0, y0, W, H, lane);
                else
                    md_build_refs(M, st, lane);
            }
            __syncthreads();
        }
        MD_PROF(2);
        if (prof_on && t == 0)
            M.prof[13] += (unsigned long long)ncand, M.prof[14] += 1, M.prof_d[M.prof_depth][13] += (unsigned long long)ncand, M.prof_d[M.prof_depth][14] += 1;
        /* ---- fast loop (ProductPerformFastLoop's second loop): ONE list of tasks = (candidate, plane, tile) dealt to the four waves ----
         * luma tasks first - a candidate of a 64x64 unit is motion-compensated in four 32x32 tiles, a task each (wave w takes tile w of EVERY candidate: all four waves work
         * whatever the number of candidates) -, then, in CHROMA_MODE_FULL LCUs, the Cb and the Cr block of every evaluated candidate.  A task predicts its block (inter:
         * ep_inter_predict_core into the candidate's slot - kept for the full loop - or the wave's scratch; intra: per sample in closed form), measures it against the source
         * (v_sad_u8 on words) and adds its part to the candidate's distortion. */
        {
            bool tiled64 = false;
            int nhc = 0;
            if constexpr (INTER) {
                tiled64 = N == 64 && !M.any_intra;
                nhc = M.V.nheavyc;
            }
            const int T = tiled64 ? 4 : 1, nl = M.nheavy * T, ntask = nl + 2 * nhc;
            for (int tk = wave; tk < ntask; tk += 4) {
                const bool luma = tk < nl;
                const int k = luma ? (tiled64 ? tk >> 2 : tk) : (tk - nl) >> 1, ti = luma ? (tiled64 ? tk & 3 : 0) : 0, pl = luma ? 0 : 1 + ((tk - nl) & 1);
                int c;
                if constexpr (INTER)
                    c = luma ? M.heavy[k] : M.V.heavyc[k];
                else
                    c = M.heavy[k];
                MD_TR(20);
                const MdCand cd = M.cand[c];
                uint32_t sad = 0;
                if (cd.type == MD_INTER) {
                    if constexpr (INTER) {
                        const int sl = M.V.slot[c], n = luma ? N : N >> 1, lgn = luma ? lgN : lgN - 1;
                        uint8_t *pr = luma ? (sl >= 0 ? M.V.cpred[sl] : M.V.wpred[wave]) : (sl >= 0 ? M.V.cpred_c[sl][pl - 1] : M.V.wpred_c(wave, pl - 1));
                        md_predict_inter_plane(M.V.refs, cd, x0, y0, N, pl, lane, M.V.mc[wave], pr, tiled64 && luma ? ti : 0, tiled64 && luma ? 4 : 1, &M.V.rw, &M.V.rwc);
                        MD_TR(22);
                        MD_SUB(7);
                        if (luma && tiled64) {
                            const int ty0 = (ti >> 1) << 5, tx0 = (ti & 1) << 5;
                            for (int e = 4 * lane; e < 32 * 32; e += 256) { /* v_sad_u8: four samples a word */
                                const int y = ty0 + (e >> 5), x = tx0 + (e & 31);
                                sad = __builtin_amdgcn_sad_u8(*reinterpret_cast<const uint32_t *>(&pr[y * 64 + x]), *reinterpret_cast<const uint32_t *>(&L.src[(st.y + y) * 64 + st.x + x]), sad);
                            }
                        } else if (luma) {
                            for (int e = 4 * lane; e < N * N; e += 256)
                                sad = __builtin_amdgcn_sad_u8(*reinterpret_cast<const uint32_t *>(&pr[e]), *reinterpret_cast<const uint32_t *>(&L.src[(st.y + (e >> lgN)) * 64 + st.x + (e & (N - 1))]), sad);
                        } else { /* chroma blocks are 4 .. 32 samples wide: rows of words */
                            const uint8_t *sc = &M.V.src_c[pl - 1][(st.y >> 1) * 32 + (st.x >> 1)];
                            for (int e = 4 * lane; e < n * n; e += 256)
                                sad = __builtin_amdgcn_sad_u8(*reinterpret_cast<const uint32_t *>(&pr[e]), *reinterpret_cast<const uint32_t *>(&sc[(e >> lgn) * 32 + (e & (n - 1))]), sad);
                        }
                    }
                } else if (luma) {
                    const int mode = cd.intra_mode;
                    sad = md_intra_block(mode, N, lgN, (!open_loop && md_mode_filtered(mode, lgN)) ? M.reff : M.ref, 1, lane, &L.src[st.y * 64 + st.x], 64, nullptr);
                } else {
                    if constexpr (INTER) { /* IntraPredictionOl's chroma pair: the chroma mode is always DM (Codec/EbIntraPrediction.c:5530); kept for the full loop like the inter ones */
                        sad = md_intra_block(cd.intra_mode, N >> 1, lgN - 1, M.V.refc[pl - 1], 0, lane, &M.V.src_c[pl - 1][(st.y >> 1) * 32 + (st.x >> 1)], 32, nullptr);
                    }
                }
                sad = md_wave_sum(sad);
                MD_TR(23);
                MD_SUB(8);
                if (lane == 0 && sad) {
                    if (luma)
                        atomicAdd(&M.sad[c], sad);
                    else if constexpr (INTER)
                        atomicAdd(&M.V.sadc[c], sad);
                }
                MD_TR(24);
            }
        }
        __syncthreads();
        MD_TR(25);
        MD_PROF(3);
        /* ---- wave 0: fast costs (a lane per candidate: MD_MAX_CAND <= 64), candidate buffers (a lane per buffer), PreModeDecision ---- */
        if (wave == 0) {
            const int i = lane;
            unsigned long long rate = 0, cst = ~0ull;
            int evl = 0;
            if (i < ncand) {
                evl = M.evaluated[i];
                if (evl) {
                    const uint64_t dist = M.cand[i].mpm ? 0 : M.sad[i];
                    uint64_t distc = 0;
                    uint32_t cw = 0;
                    if constexpr (INTER) {
                        if (M.lcu.chroma_encode_mode == 1) /* the chroma pair's SAD with the noise-class rule (:2079-2094) */
                            distc = M.cand[i].mpm ? 0 : md_fast_chroma_noise_rule(&M.lcu, N, &M.cand[i], M.V.sadc[i]), cw = M.V.X.chroma_weight;
                    }
                    cst = M.cand[i].type == MD_INTER ? md_inter_fast_cost_c(&P, &st, &M.S.cu[leaf], &M.cand[i], dist, distc, cw, !M.lcu.cmplx_noise, (uint64_t *)&rate)
                          : islice                  ? md_intra_fast_cost_islice(&P, &st, &M.S.cu[leaf], M.cand[i].intra_mode, dist, (uint64_t *)&rate)
                                                    : md_intra_fast_cost_pslice_c(&P, &st, &M.S.cu[leaf], M.cand[i].intra_mode, dist, distc, cw, (uint64_t *)&rate);
                    if (M.cand[i].mpm)
                        cst = 0;
                }
                M.costs[i] = cst, M.fast_rate[i] = rate;
            }
            MD_TR(26);
            MD_SUB(9);
            /* md_fast_loop_buffers (md_logic.h; ProductPerformFastLoop's second loop, :1990-2179) with the buffers in lanes 0..7 instead of LDS: the candidates
             * arrive from the last to the first, each goes into the buffer with the highest cost (an unused one first) = the FIRST buffer holding the maximum
             * over [0, maxBuffers) - the reference's scan starts at buffer 0, moves on a strictly greater cost and stops at an unused (all-ones) one; its
             * do-while looks at buffer 1 even when maxBuffers is 1.  Scalar, the replay of 35 intra candidates was a third of an I picture's time. */
            unsigned long long bcost = ~0ull;
            int bcand = -1, bpred = -1, evcount = 0, highest = 0;
            const int maxb = __builtin_amdgcn_readfirstlane(M.max_buffers < 2 ? 2 : M.max_buffers);
            for (int idx = __builtin_amdgcn_readfirstlane(ncand) - 1; idx >= 0; idx--) {
                const unsigned long long c = md_readlane64(cst, idx);
                const int ev = __builtin_amdgcn_readlane(evl, idx);
                if (lane == highest) {
                    bcand = idx;
                    if (ev) {
                        bcost = c;
                        if (!(ev & 2))
                            bpred = idx;
                    }
                }
                evcount += ev != 0;
                if (idx) { /* the first buffer holding the maximum: the buffers' costs read lane by lane (scalar) */
                    unsigned long long m = 0;
                    int h = 0;
#pragma unroll
                    for (int b = 0; b < MD_MAX_BUF; b++)
                        if (b < maxb) {
                            const unsigned long long v = md_readlane64(bcost, b);
                            if (b == 0 || v > m)
                                m = v, h = b;
                        }
                    highest = h;
                }
            }
            MD_TR(27);
            MD_SUB(10);
            if (lane < MD_MAX_BUF) {
                M.B.fast_cost[lane] = bcost, M.B.full_cost[lane] = ~0ull, M.B.cand[lane] = (int16_t)bcand, M.B.pred[lane] = (int16_t)bpred;
                M.types[lane] = bcand >= 0 ? M.cand[bcand].type : 0, M.ycbf[lane] = 0, M.full_dist[lane] = 0, M.merge_cost[lane] = M.skip_cost[lane] = 0;
                M.y_bits[lane] = M.y_dist[lane][0] = M.y_dist[lane][1] = 0;
            }
            if (lane == 0)
                M.B.evaluated_count = evcount;
            EP_WAVE_SYNC();
            if (lane == 0) {
                int bufferTotal = M.buffer_total;
                bufferTotal = evcount < bufferTotal ? evcount : bufferTotal;
                const int same = evcount == bufferTotal;
                M.full_count = md_pre_mode_decision(&M.B, M.types, same ? bufferTotal : M.max_buffers, same, M.best);
                M.nfull = M.full_count < bufferTotal ? M.full_count : bufferTotal;
            }
            MD_TR(28);
        }
        __syncthreads();
        MD_TR(29);
        MD_PROF(4);
        /* ---- full loop: a wave per surviving candidate (PerformFullLoop, :4351) ---- */
        const int nfull = M.nfull;
        bool split64 = false, fresh64 = false;
        if constexpr (INTER) {
            /* a 64x64 unit has four 32x32 transform units per candidate: a wave per (candidate, transform unit) instead of a wave per candidate - with the usual one or
             * two survivors all four waves work.  The candidates of a 64x64 unit are motion-compensated (no intra candidate at depth 0). */
            split64 = N == 64 && nfull <= 4 && !M.any_intra;
            if (split64) {
                bool sync = false;
                for (int f = 0; f < nfull; f++) {
                    const int ci = M.B.cand[M.best[f]], pci = M.B.pred[M.best[f]] < 0 ? ci : M.B.pred[M.best[f]];
                    if (!(M.V.slot[pci] >= 0 && M.evaluated[pci])) {
                        sync = true;
                        if (wave == f)
                            md_predict_inter_plane(M.V.refs, M.cand[pci], x0, y0, N, 0, lane, M.V.mc[wave], M.V.wpred[f], 0, 1, &M.V.rw, &M.V.rwc);
                    }
                }
                if (sync)
                    __syncthreads();
                fresh64 = sync;
                for (int f = 0; f < nfull; f++) {
                    const int b = M.best[f], ci = M.B.cand[b], pci = M.B.pred[b] < 0 ? ci : M.B.pred[b];
                    const uint8_t *pred = (M.V.slot[pci] >= 0 && M.evaluated[pci]) ? M.V.cpred[M.V.slot[pci]] : M.V.wpred[f];
                    const int tu = wave, off = ((tu & 1) << 5) + ((tu >> 1) << 5) * 64;
                    const MdFl o = md_full_loop_unit<32>(lane, &L.src[st.y * 64 + st.x] + off, 64, pred + off, 64, nullptr, M.tiles[wave], M.qbuf[wave], P.qp, P.slice_type, M.cost,
                                                         M.cand[ci].type, M.cand[ci].intra_mode, 0, pf, M.rt);
                    if (lane == 0)
                        M.fl[b][tu] = o;
                    EP_WAVE_SYNC();
                }
            }
        }
        for (int f = wave; f < nfull && !split64; f += 4) {
            MD_TR(30);
            const int b = M.best[f], ci = M.B.cand[b];
            const MdCand cd = M.cand[ci];
            /* the buffer's luma prediction: the candidate the fast loop predicted there, or - predictionIsReadyLuma == 0 - a fresh one */
            const bool fresh = ci == M.best_first && cd.type == MD_INTRA && open_loop && M.evaluated[ci];
            const int pci = (fresh || M.B.pred[b] < 0) ? ci : M.B.pred[b];
            const MdCand pc = M.cand[pci];
            uint8_t *pred;
            int16_t *rc = nullptr;
            if constexpr (INTER) {
                pred = M.V.wpred[wave];
            } else {
                pred = M.V.pred[b], rc = M.V.recon_coeff[b];
            }
            if (pc.type == MD_INTER) {
                if constexpr (INTER) {
                    if (M.V.slot[pci] >= 0 && M.evaluated[pci])
                        pred = M.V.cpred[M.V.slot[pci]]; /* the fast loop's prediction of this candidate is still there */
                    else
                        md_predict_inter_plane(M.V.refs, pc, x0, y0, N, 0, lane, M.V.mc[wave], pred, 0, 1, &M.V.rw, &M.V.rwc);
                }
            } else {
                const int mode = pc.intra_mode;
                md_intra_block(mode, N, lgN, (!open_loop && md_mode_filtered(mode, lgN)) ? M.reff : M.ref, 1, lane, nullptr, 0, pred);
                EP_WAVE_SYNC();
            }
            MD_TR(31);
            MD_SUB(12);
            md_full_loop_cand(lane, N, &L.src[st.y * 64 + st.x], pred, N, rc, M.tiles[wave], M.qbuf[wave], P, M.cost, M.rt, cd.type, cd.intra_mode, pf, M.fl[b]);
            MD_TR(32);
            MD_SUB(13);
        }
        if constexpr (INTER) {
            /* CHROMA_MODE_FULL (PerformFullLoop :4443-4560): the chroma pair of every survivor - ChromaPrediction (the candidate's OWN prediction: the fast loop's when it
             * evaluated the candidate, a fresh one otherwise), FullLoop_R + CuFullDistortionFastTuMode_R - as tasks (survivor, plane) on the waves the luma units left idle */
            if (M.lcu.chroma_encode_mode == 1) {
                if (fresh64) /* every wave is done with the fresh luma predictions in the waves' scratch before a chroma block lands there */
                    __syncthreads();
                const int Cn = N >> 1, lgc = lgN - 1, Tc = N == 64 ? 16 : Cn, ntu = N == 64 ? 4 : 1;
                /* which wave takes which (survivor, plane) task: the waves that carried a luma unit start with 3 units of load, a chroma pair member costs 2 - each task
                 * goes to the least loaded wave (every wave derives the same table; two survivors: both chroma pairs on the two waves the luma units left idle instead
                 * of one member behind each luma unit) */
                unsigned mine = 0;
                {
                    int load[4];
#pragma unroll
                    for (int w_ = 0; w_ < 4; w_++)
                        load[w_] = (!split64 && w_ < nfull) ? 3 : 0;
                    for (int tk = 0; tk < 2 * nfull; tk++) {
                        int best_w = 0;
#pragma unroll
                        for (int w_ = 1; w_ < 4; w_++)
                            if (load[w_] < load[best_w])
                                best_w = w_;
#pragma unroll
                        for (int w_ = 0; w_ < 4; w_++)
                            if (w_ == best_w)
                                load[w_] += 2;
                        if (best_w == wave)
                            mine |= 1u << tk;
                    }
                }
                for (int tk = 0; tk < 2 * nfull; tk++) {
                    if (!((mine >> tk) & 1u))
                        continue;
                    const int f = tk >> 1, pl = tk & 1, b = M.best[f], ci = M.B.cand[b];
                    const MdCand cd = M.cand[ci];
                    const uint8_t *pred;
                    if (cd.type == MD_INTER && M.V.slot[ci] >= 0 && M.evaluated[ci]) {
                        pred = M.V.cpred_c[M.V.slot[ci]][pl];
                    } else {
                        uint8_t *pw = M.V.wpred_c(wave, pl);
                        if (cd.type == MD_INTER) {
                            md_predict_inter_plane(M.V.refs, cd, x0, y0, N, 1 + pl, lane, M.V.mc[wave], pw, 0, 1, &M.V.rw, &M.V.rwc);
                        } else {
                            md_intra_block(cd.intra_mode, Cn, lgc, M.V.refc[pl], 0, lane, nullptr, 0, pw);
                            EP_WAVE_SYNC();
                        }
                        pred = pw;
                    }
                    for (int tu = 0; tu < ntu; tu++) {
                        const int ox = ntu == 1 ? 0 : (tu & 1) << 4, oy = ntu == 1 ? 0 : (tu >> 1) << 4;
                        uint32_t nz;
                        unsigned long long d[2], bt;
                        md_chroma_tu(lane, Tc, &M.V.src_c[pl][((st.y >> 1) + oy) * 32 + (st.x >> 1) + ox], pred + oy * Cn + ox, Cn, M.tiles[wave], M.qbuf[wave], P, M.cost, M.rt, cd.type,
                                     cd.intra_mode, 1 + pl, pf, &nz, d, &bt);
                        if (lane == 0) {
                            MdFl o;
                            o.nz = nz, o.d0 = (uint32_t)d[0], o.d1 = (uint32_t)d[1], o.bits = (uint32_t)bt;
                            M.V.flc[b][pl][tu] = o;
                        }
                    }
                }
            }
        }
        MD_TR(33);
        __syncthreads();
        MD_TR(34);
        MD_PROF(5);
        /* ---- wave 0: TuCalcCostLuma + the full cost of every surviving candidate (a lane each), then lane 0: ProductFullModeDecision, CheckHighCostPartition ---- */
        if (wave == 0) {
            const bool have = lane < nfull;
            int b = 0, ctype = 0;
            uint32_t ycbf = 0;
            unsigned long long bits = 0, dist[2] = {0, 0}, full = 0;
            uint64_t mc = 0, sc = 0;
            if (have) {
                const SvtAmdMdPicture &PL = M.pic; /* the rate tables beside the LCU */
                b = M.best[lane];
                const int ci = M.B.cand[b];
                const MdCand c = M.cand[ci];
                ctype = c.type;
                if (N == 64) {
                    for (int tu = 0; tu < 4; tu++)
                        md_tu_calc_cost(PL, M.fl[b][tu], c.type, 64, 32, tu + 1, &ycbf, &bits, dist);
                } else {
                    md_tu_calc_cost(PL, M.fl[b][0], c.type, N, N, 0, &ycbf, &bits, dist);
                }
                if (M.lcu.chroma_encode_mode == 2 /* CHROMA_MODE_BEST */ || M.lcu.chroma_encode_mode == 1)
                    bits = md_pf_coeff_bits(pf, P.qp, bits);
                bool with_chroma = false;
                if constexpr (INTER) {
                    if (M.lcu.chroma_encode_mode == 1) { /* InterFullCost / MergeSkipFullCost / IntraFullCostPslice: the chroma loop's sums join the luma ones */
                        with_chroma = true;
                        const int ntu = N == 64 ? 4 : 1;
                        uint32_t cbf[2] = {0, 0};
                        uint64_t cbits[2] = {0, 0}, cdist[2][2] = {{0, 0}, {0, 0}};
                        for (int pl = 0; pl < 2; pl++)
                            for (int tu = 0; tu < ntu; tu++) {
                                const MdFl o = M.V.flc[b][pl][tu];
                                cbf[pl] |= (uint32_t)(o.nz != 0) << (ntu == 1 ? 0 : tu + 1);
                                cbits[pl] += o.bits, cdist[pl][0] += o.d0, cdist[pl][1] += o.d1;
                            }
                        const uint64_t yd[2] = {dist[0], dist[1]};
                        if (c.type == MD_INTER)
                            full = md_inter_full_cost(&PL, M.V.X.chroma_weight, &M.S.cu[leaf], &c, N, ycbf, cbf, M.fast_rate[ci], yd, cdist, bits, cbits, &mc, &sc);
                        else
                            full = md_intra_full_cost_pslice(&PL, M.V.X.chroma_weight, N, ycbf, cbf, M.fast_rate[ci], dist[0], cdist, bits, cbits);
                    }
                }
                if (with_chroma)
                    ;
                else if (c.type == MD_INTER)
                    full = md_inter_full_luma_cost(&PL, &M.S.cu[leaf], &c, N, ycbf, M.fast_rate[ci], (const uint64_t *)dist, bits, &mc, &sc);
                else if (islice)
                    full = md_intra_full_luma_cost_islice(&PL, lgN, ycbf, M.fast_rate[ci], dist[0], bits);
                else
                    full = md_intra_full_luma_cost_pslice(&PL, N, ycbf, M.fast_rate[ci], dist[0], bits);
            }
            /* the reference walks the candidates in order: an intra candidate after an inter one whose root cbf is 0 is not costed at all (full-loop escape, :4450-4460) and
             * keeps whatever its buffer held */
            uint32_t prevRootCbf = 1;
            unsigned long long bestFullCost = 0xFFFFFFFFull, kept = 0;
            for (int g = 0; g < nfull; g++) {
                const int ty = __shfl(ctype, g);
                const uint32_t yc = __shfl(ycbf, g);
                const unsigned long long cs = __shfl(full, g);
                if (!islice && ty == MD_INTRA && prevRootCbf == 0)
                    continue;
                kept |= 1ull << g;
                if (P.full_loop_escape && !islice && ty == MD_INTER && cs < bestFullCost)
                    prevRootCbf = yc, bestFullCost = cs;
            }
            MD_TR(35);
            MD_SUB(14);
            if (have && ((kept >> lane) & 1ull)) {
                M.ycbf[b] = ycbf, M.full_dist[b] = (uint32_t)dist[0], M.B.full_cost[b] = full;
                M.merge_cost[b] = mc, M.skip_cost[b] = sc, M.y_bits[b] = bits, M.y_dist[b][0] = dist[0], M.y_dist[b][1] = dist[1];
            }
            EP_WAVE_SYNC();
        }
        if (t == 0) {
            int lowest = M.best[0];
            unsigned long long lowestCost = ~0ull;
            for (int f = 0; f < M.full_count; f++)
                if (M.B.full_cost[M.best[f]] < lowestCost)
                    lowest = M.best[f], lowestCost = M.B.full_cost[M.best[f]];
            if (ncand > 0) {
                const MdCand c = M.cand[M.B.cand[lowest]];
                MdCu &u = M.S.cu[leaf];
                M.S.local[leaf].cost = M.B.full_cost[lowest], M.S.local[leaf].full_distortion = M.full_dist[lowest];
                u.pred_mode = c.type, u.skip_flag = 0, u.intra_luma_mode = (uint8_t)(c.type == MD_INTRA ? c.intra_mode : 0x1F);
                u.ycbf = (uint8_t)(N == 64 ? (M.ycbf[lowest] & 0x1E) : (M.ycbf[lowest] & 1));
                u.inter_dir = (uint8_t)(c.type == MD_INTER ? c.dir : 3), u.merge_flag = (uint8_t)(c.type == MD_INTER ? c.merge_flag : 0), u.merge_index = c.merge_index;
                u.mv[0].x = u.mv[0].y = u.mv[1].x = u.mv[1].y = 0;
                if (c.type == MD_INTER) {
                    if (c.dir != MD_L1)
                        u.mv[0] = c.mv[0];
                    if (c.dir != MD_L0)
                        u.mv[1] = c.mv[1];
                }
                u.merge_cost = M.merge_cost[lowest], u.skip_cost = M.skip_cost[lowest];
                u.y_coeff_bits = M.y_bits[lowest], u.y_dist[0] = M.y_dist[lowest][0], u.y_dist[1] = M.y_dist[lowest][1];
                u.fast_luma_rate = M.fast_rate[M.B.cand[lowest]], u.ycbf_mask = M.ycbf[lowest];
            }
            M.lowest = lowest;
            M.S.local[leaf].mdc_index = (uint8_t)M.cu_idx;
            const int exitParent = md_check_high_cost_partition(&P, &M.lcu, &M.S, leaf);
            M.do_recon = exitParent < 0 && ncand > 0 && !open_loop, M.exited = exitParent >= 0;
            if (exitParent >= 0) {
                M.leaf = exitParent, M.cu_idx = M.S.local[exitParent].mdc_index;
                M.S.cu[exitParent].split = 0;
                M.last = md_inter_depth_decision(&P, &M.S, exitParent, lcu_x, lcu_y, 1, 0);
            } else if (open_loop) { /* no reconstruction to wait for: the inter-depth decision follows at once */
                M.last = md_inter_depth_decision(&P, &M.S, leaf, lcu_x, lcu_y, 0, md_stop_split(&P, &M.lcu, st.depth, M.S.local[leaf].full_distortion));
            }
            if (exitParent >= 0 || open_loop)
                M.update = M.S.cu[M.last].split == 0;
            MD_TR(36);
        }
        __syncthreads();
        MD_TR(37);
        MD_PROF(6);
        if constexpr (!INTER) {
            /* ---- wave 0: the winner's reconstruction ---- */
            if (M.do_recon && wave == 0) {
                const int b = M.lowest;
                uint8_t *dst = M.V.best_rec[st.depth] + st.y * 64 + st.x;
                if (M.S.cu[leaf].ycbf) {
                    switch (N) {
                    case 32: md_recon_unit<32>(lane, M.V.recon_coeff[b], M.V.pred[b], dst, M.tiles[0]); break;
                    case 16: md_recon_unit<16>(lane, M.V.recon_coeff[b], M.V.pred[b], dst, M.tiles[0]); break;
                    default: md_recon_unit<8>(lane, M.V.recon_coeff[b], M.V.pred[b], dst, M.tiles[0]); break;
                    }
                } else {
                    for (int e = lane; e < N * N; e += 64)
                        dst[(e >> lgN) * 64 + (e & (N - 1))] = M.V.pred[b][e];
                }
            }
            __syncthreads();
            /* ---- lane 0: inter-depth decision ---- */
            if (t == 0 && !open_loop) {
                if (!M.exited)
                    M.last = md_inter_depth_decision(&P, &M.S, leaf, lcu_x, lcu_y, 0, md_stop_split(&P, &M.lcu, st.depth, M.S.local[leaf].full_distortion));
                M.update = M.S.cu[M.last].split == 0;
            }
            __syncthreads();
            MD_PROF(7);
        }
        /* ---- all lanes: ModeDecisionUpdateNeighborArrays of the unit the decision ended on ---- */
        if (M.update) {
            const int last = M.last;
            const MdStats ls = md_stats(last);
            const MdCu u = M.S.cu[last];
            const uint32_t w = (uint32_t)u.pred_mode | ((uint32_t)u.intra_luma_mode << 8) | ((uint32_t)ls.depth << 16) | ((uint32_t)u.skip_flag << 24);
            if constexpr (!INTER) {
                if (!open_loop) {
                    const uint8_t *srcp = M.V.best_rec[ls.depth];
                    for (int e = t; e < ls.size * ls.size; e += 256) {
                        const int y = e >> ls.lg, x = e & (ls.size - 1);
                        *L.at(ls.x + x, ls.y + y) = srcp[(ls.y + y) * 64 + ls.x + x];
                    }
                }
            }
            const int cells = ls.size >> 2;
            for (int e = t; e < cells * cells; e += 256)
                L.info[((ls.y >> 2) + e / cells + 1) * 36 + (ls.x >> 2) + e % cells + 1] = w;
            if constexpr (INTER) {
                const int c8 = ls.size >> 3;
                MdMvUnit mu;
                mu.mv[0] = u.mv[0], mu.mv[1] = u.mv[1], mu.dir = u.inter_dir, mu.avail = 0, mu.pad[0] = mu.pad[1] = 0;
                for (int e = t; e < c8 * c8; e += 256)
                    M.V.mvu[((ls.y >> 3) + e / c8 + 1) * 18 + (ls.x >> 3) + e % c8 + 1] = mu;
            }
        }
        MD_TR(38);
        __syncthreads();
        MD_TR(39);
        if (t == 0) {
            const int cur = M.leaf; /* the unit the loop stands on: the tested one, or the parent a partition exit fell back to */
            const MdStats cs = md_stats(cur);
            int cuIdx = M.cu_idx;
            if (M.S.cu[cur].split)
                cuIdx++;
            else if (lh < 64)
                cuIdx++;
            else
                cuIdx += md_next_cu_step(&M.lcu, cuIdx, cs.depth);
            M.cu_idx = cuIdx;
            M.done = cuIdx >= M.lcu.leaf_count;
#ifdef MD_TRACE
            g_md_trace_on = D.trace && lcu == D.trace_lcu && cuIdx >= D.trace_unit && cuIdx < D.trace_unit + 2;
            if (D.trace && lcu == D.trace_lcu)
                D.trace[4 * (1 + MD_TRACE_N) + 3] += 0x10000ull + (g_md_trace_on ? 1 : 0) + ((unsigned long long)g_md_trace_n[0] << 32);
#endif
        }
        __syncthreads();
        MD_PROF(8);
        if (M.done)
            break;
    }
    /* ---- the LCU's state leaves LDS: neighbour maps of the picture + the decision record ---- */
    if (!INTER) {
        for (int i = t; i < 64 * 64 / 4; i += 256) {
            const int y = i >> 4, x = (i & 15) * 4;
            if (x < lw && y < lh)
                *(uint32_t *)(D.md_rec + (size_t)(lcu_y + y) * D.md_pitch + lcu_x + x) = *(const uint32_t *)L.at(x, y);
        }
    }
    for (int i = t; i < 16 * 16; i += 256) {
        const int cy = i >> 4, cx = i & 15;
        if (4 * cx < lw && 4 * cy < lh)
            D.md_info[(size_t)((lcu_y >> 2) + cy) * D.info_pitch + (lcu_x >> 2) + cx] = L.info[(cy + 1) * 36 + cx + 1];
    }
    if constexpr (INTER) {
        for (int i = t; i < 8 * 8; i += 256) {
            const int cy = i >> 3, cx = i & 7;
            if (8 * cx < lw && 8 * cy < lh) {
                const MdMvUnit u = M.V.mvu[(cy + 1) * 18 + cx + 1];
                uint4 w;
                w.x = (uint32_t)(uint16_t)u.mv[0].x | ((uint32_t)(uint16_t)u.mv[0].y << 16), w.y = (uint32_t)(uint16_t)u.mv[1].x | ((uint32_t)(uint16_t)u.mv[1].y << 16);
                w.z = u.dir, w.w = 0;
                D.md_mv[(size_t)((lcu_y >> 3) + cy) * D.mv_pitch + (lcu_x >> 3) + cx] = w;
            }
        }
    }
    if (D.out) {
        SvtAmdMdLcuOut &O = D.out[lcu];
        for (int i = t; i < SVT_AMD_MD_LEAVES; i += 256) {
            const MdCu u = M.S.cu[i];
            O.split[i] = u.split, O.tested[i] = M.S.local[i].tested, O.pred_mode[i] = u.pred_mode;
            O.intra_luma_mode[i] = u.intra_luma_mode, O.ycbf[i] = u.ycbf, O.cost[i] = M.S.local[i].cost;
            O.inter_dir[i] = u.inter_dir, O.merge_flag[i] = u.merge_flag, O.merge_index[i] = u.merge_index;
            O.mv[i][0][0] = u.mv[0].x, O.mv[i][0][1] = u.mv[0].y, O.mv[i][1][0] = u.mv[1].x, O.mv[i][1][1] = u.mv[1].y;
            O.merge_cost[i] = u.merge_cost, O.skip_cost[i] = u.skip_cost;
        }
    }
}

/* what EncodePass will do with the inter units of the LCU's final tree (Codec/EbCodingLoop.c:3838-3882): AMVP units as they are; merge units by the
 * merge / skip costs completed with chroma (AddChromaEncDec, Codec/EbProductCodingLoop.c:4158-4349: chroma prediction + chroma full loop +
 * MergeSkipFullCost), a wave per unit.  -> M.V.ep_kind[leaf] */
__device__ __forceinline__ void md_ep_kinds(const MdPictureDev &D, const SvtAmdMdPicture &Pg, MdShared<true> &M, int lcu_x, int lcu_y)
{
    const SvtAmdMdPicture &P = M.pic;
    (void)Pg;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int lw = min(64, (int)P.width - lcu_x), lh = min(64, (int)P.height - lcu_y);
    if (t == 0) {
        int n = 0, it = 0;
        while (it < SVT_AMD_MD_LEAVES) {
            if (M.S.cu[it].split) {
                it++;
                continue;
            }
            const MdStats st = md_stats(it);
            if (lcu_x + st.x < (int)P.width && lcu_y + st.y < (int)P.height && M.S.cu[it].pred_mode == MD_INTER) {
                M.V.ep_kind[it] = M.S.cu[it].merge_flag ? SVT_AMD_EP_INTER_MERGE : SVT_AMD_EP_INTER_AMVP;
                if (M.S.cu[it].merge_flag && M.lcu.chroma_encode_mode == 1) /* CHROMA_MODE_FULL: the mode decision's merge / skip costs hold chroma already (EbCodingLoop.c:3840) */
                    M.V.ep_kind[it] = (uint8_t)md_ep_merge_kind(&M.V.X, &M.lcu, M.S.cu[it].merge_cost, M.S.cu[it].skip_cost);
                else if (M.S.cu[it].merge_flag && n < SVT_AMD_LCU_MAX_CUS)
                    M.V.fin_leaf[n++] = (uint8_t)it;
            }
            it += md_depth_offset(st.depth);
        }
        M.V.nfin = n;
    }
    __syncthreads(); /* (the LCU's chroma source is in LDS since md_lcu_inputs) */
    const int pf = md_pf_mode(&P);
    for (int i = wave; i < M.V.nfin; i += 4) {
        const int leaf = M.V.fin_leaf[i];
        const MdStats st = md_stats(leaf);
        const MdCu u = M.S.cu[leaf];
        const int N = st.size, Cn = N >> 1, T = N == 64 ? 16 : Cn, ntu = N == 64 ? 4 : 1;
        uint32_t cbf[2] = {0, 0};
        uint64_t bits[2] = {0, 0}, dist[2][2] = {{0, 0}, {0, 0}};
        MdCand cd; /* the unit's final prediction as a candidate: from the reference pictures as the mode decision reads them (their 8-MSB views of a 10-bit picture) */
        cd.type = MD_INTER, cd.dir = u.inter_dir, cd.mv[0] = u.mv[0], cd.mv[1] = u.mv[1];
        for (int p = 0; p < 2; p++) {
            uint8_t *pred = M.V.wpred[wave] + p * 1024;
            md_predict_inter_plane(M.V.refs, cd, lcu_x + st.x, lcu_y + st.y, N, 1 + p, lane, M.V.mc[wave], pred, 0, 1, &M.V.rw, &M.V.rwc);
            for (int tu = 0; tu < ntu; tu++) {
                const int ox = ntu == 1 ? 0 : (tu & 1) << 4, oy = ntu == 1 ? 0 : (tu >> 1) << 4;
                uint32_t nz;
                unsigned long long d[2], b;
                md_chroma_tu(lane, T, &M.V.src_c[p][((st.y >> 1) + oy) * 32 + (st.x >> 1) + ox], pred + oy * Cn + ox, Cn, M.tiles[wave], M.qbuf[wave], P, M.cost, M.rt, MD_INTER, 0,
                             1 + p, pf, &nz, d, &b);
                cbf[p] |= (uint32_t)(nz != 0) << (ntu == 1 ? 0 : tu + 1);
                bits[p] += b, dist[p][0] += d[0], dist[p][1] += d[1];
            }
        }
        if (lane == 0) {
            uint64_t mc, sc;
            md_merge_skip_full_cost(&P, &M.V.X, &u, N, cbf, bits, dist, &mc, &sc);
            M.V.ep_kind[leaf] = (uint8_t)md_ep_merge_kind(&M.V.X, &M.lcu, mc, sc);
        }
    }
    __syncthreads();
}

/* the EncDec input contract the decisions amount to (what svt_hook_encdec.c:fill_work builds on the host): the final tree in Z order */
template <bool INTER, typename T>
__device__ __forceinline__ void md_make_work(const MdPictureDev &D, const SvtAmdMdPicture &P, const MdShared<INTER> &M, int lcu_x, int lcu_y, typename EpTypes<T>::Work &Wk)
{
    const int t = threadIdx.x;
    const int lw = min(64, (int)P.width - lcu_x), lh = min(64, (int)P.height - lcu_y);
    if (t == 0) {
        Wk.lcu_x = (uint16_t)lcu_x, Wk.lcu_y = (uint16_t)lcu_y;
        Wk.slice_type = P.slice_type, Wk.temporal_layer = P.temporal_layer, Wk.constrained_intra = P.constrained_intra, Wk.strong_smoothing = P.strong_smoothing;
        Wk.tile_left = M.lcu.tile_left, Wk.tile_top = M.lcu.tile_top, Wk.tile_right = M.lcu.tile_right;
        Wk.full_lambda = P.full_lambda;
        Wk.luma_cbf_bits[0] = P.rates.lumaCbfBits[0], Wk.luma_cbf_bits[1] = P.rates.lumaCbfBits[1];
        Wk.luma_cbf_bits[2] = P.rates.lumaCbfBits[5], Wk.luma_cbf_bits[3] = P.rates.lumaCbfBits[6];
        Wk.pm_core = 0;
        int n = 0, it = 0;
        while (it < SVT_AMD_MD_LEAVES) {
            if (M.S.cu[it].split) {
                it++;
                continue;
            }
            const MdStats st = md_stats(it);
            if (lcu_x + st.x < (int)P.width && lcu_y + st.y < (int)P.height && n < SVT_AMD_LCU_MAX_CUS) {
                SvtAmdLcuCu &u = Wk.cu[n++];
                const MdCu &c = M.S.cu[it];
                u.x = st.x, u.y = st.y, u.size = st.size, u.pred_mode = c.pred_mode, u.intra_luma_mode = c.pred_mode == MD_INTRA ? c.intra_luma_mode : 0;
                u.bottom_left_ok = (uint8_t)md_bottom_left_ok(&st), u.top_right_ok = (uint8_t)md_top_right_ok(&st);
                u.qp = P.qp, u.chroma_qp = P.chroma_qp, u.leaf_index = (uint8_t)it, u.inter_dir = 0, u.inter_kind = 0, u.dz_offset = 0;
                u.mv[0][0] = u.mv[0][1] = u.mv[1][0] = u.mv[1][1] = 0;
                if (c.pred_mode == MD_INTER) {
                    u.inter_dir = c.inter_dir;
                    if constexpr (INTER)
                        u.inter_kind = M.V.ep_kind[it];
                    u.mv[0][0] = c.mv[0].x, u.mv[0][1] = c.mv[0].y, u.mv[1][0] = c.mv[1].x, u.mv[1][1] = c.mv[1].y;
                }
            }
            it += md_depth_offset(st.depth);
        }
        Wk.num_cus = (uint8_t)n;
    }
    if constexpr (sizeof(T) == 1) {
        for (int i = t; i < 64 * 64 / 4; i += 256)
            ((uint32_t *)Wk.src_y)[i] = ((const uint32_t *)M.L.src)[i];
        for (int i = t; i < 2 * 32 * 32; i += 256) {
            const int p = i >> 10, e = i & 1023, y = e >> 5, x = e & 31;
            uint8_t v = 0;
            if (x < lw / 2 && y < lh / 2)
                v = D.src[1 + p][(size_t)(lcu_y / 2 + y) * D.src_pitch[1] + lcu_x / 2 + x];
            (p ? Wk.src_cr : Wk.src_cb)[e] = v;
        }
    } else { /* the encode pass of a 10-bit picture codes the 10-bit source (EncodePassPackLcu's inputSample16bitBuffer, EbCodingLoop.c:2867); zero outside the picture */
        for (int i = t; i < 64 * 64; i += 256) {
            const int y = i >> 6, x = i & 63;
            Wk.src_y[i] = (x < lw && y < lh) ? D.src16[0][(size_t)(lcu_y + y) * D.src_pitch[0] + lcu_x + x] : (uint16_t)0;
        }
        for (int i = t; i < 2 * 32 * 32; i += 256) {
            const int p = i >> 10, e = i & 1023, y = e >> 5, x = e & 31;
            uint16_t v = 0;
            if (x < lw / 2 && y < lh / 2)
                v = D.src16[1 + p][(size_t)(lcu_y / 2 + y) * D.src_pitch[1] + lcu_x / 2 + x];
            (p ? Wk.src_cr : Wk.src_cb)[e] = v;
        }
    }
}

/* ONE launch per picture: persistent workgroups draw LCUs as tickets in wavefront order (k_encode_picture's scheme, encdec_kernels.hip).  md_done[lcu] = the LCU's
 * mode-decision state is in the picture's maps: what the mode decision of the right and the lower-left LCU waits for.  Behind it, off the chain, the workgroup completes
 * the merge / skip decisions with chroma and writes the LCU's work record; the ENCODE PASS of the picture is a kernel of its own (k_encode_picture, launched behind this
 * one on the same stream: round 6 - inlined here it cost this kernel 904 B of private segment per lane, 43 K of its 64 K instructions and a quarter of every workgroup's
 * time, for work that is off the picture's critical path and runs 2040 LCUs wide in ~1.5 ms when it is launched on its own).
 * The picture's descriptor (MdPictureDev) comes by POINTER and is copied to LDS once per workgroup: passed by value, its run-time-indexed arrays (src[1 + p], mref[l])
 * forced the whole record into the private segment (110 scratch stores in the prologue, a memory round trip at every use). */
template <bool INTER, typename T, bool PROF>
__global__ __launch_bounds__(256) void k_md_picture(const MdPictureDev *__restrict__ Dp, typename EpTypes<T>::Work *__restrict__ works, int nlcu, int wl, unsigned *ticket,
                                                    unsigned *md_done, const unsigned *__restrict__ order, unsigned epoch)
{
    /* the LCU state as STATIC shared memory (its size is a compile-time constant): with `extern __shared__` every leaf function looked the dynamic segment's offset up
     * in a table in memory at its entry (llvm.amdgcn.dynlds.offset.table: a scalar load and its latency per call), and no access had an absolute address */
    __shared__ MdShared<INTER> M;
    __shared__ unsigned s_ticket;
    __shared__ MdPictureDev s_D;
    static_assert(sizeof(MdPictureDev) % 8 == 0, "copied as 8-byte words");
    for (int i = threadIdx.x; i < (int)(sizeof(MdPictureDev) / 8); i += 256)
        reinterpret_cast<unsigned long long *>(&s_D)[i] = reinterpret_cast<const unsigned long long *>(Dp)[i];
    __syncthreads();
    md_dct_operands_init(Dp->force_butterflies);
    const MdPictureDev &D = s_D;
    const SvtAmdMdPicture &P = *D.P;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0)
            s_ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        if ((int)s_ticket >= nlcu)
            return;
        const int lcu = (int)order[s_ticket];
        const int lx = lcu % wl, ly = lcu / wl;
        const SvtAmdMdLcu &Lc = D.lcus[lcu];
        unsigned long long c_ticket = 0;
        md_lcu_inputs<INTER>(D, P, lcu, lx * 64, ly * 64, M);
        const int dep0 = Lc.tile_left ? -1 : lcu - 1;
        const int dep1 = Lc.tile_top ? -1 : (Lc.tile_right || lx + 1 >= wl) ? lcu - wl : lcu - wl + 1;
        if (threadIdx.x == 0) {
            c_ticket = (PROF && D.prof) ? __builtin_readcyclecounter() : 0;
            if (dep0 >= 0)
                while (__hip_atomic_load(&md_done[dep0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch)
                    __builtin_amdgcn_s_sleep(2);
            if (dep1 >= 0)
                while (__hip_atomic_load(&md_done[dep1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch)
                    __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        unsigned long long c_wait = 0, c_md = 0;
        if (PROF && D.prof && threadIdx.x == 0) {
            c_wait = __builtin_readcyclecounter();
            for (int k = 0; k < 32; k++)
                M.prof[k] = 0;
            for (int k = 0; k < 128; k++)
                M.prof_d[0][k] = 0;
            M.prof_depth = 0;
            M.prof_t = M.prof_s = c_wait;
        }
#ifdef MD_TRACE
        if (threadIdx.x < 4)
            g_md_trace_n[threadIdx.x] = 0;
        if (threadIdx.x == 0)
            g_md_trace_on = D.trace && lcu == D.trace_lcu && D.trace_unit == 0;
        __syncthreads();
#endif
        md_lcu<INTER, PROF>(D, P, lcu, lx * 64, ly * 64, M);
        __syncthreads();
#ifdef MD_TRACE
        if (D.trace && lcu == D.trace_lcu) {
            for (int i = threadIdx.x; i < 4 * (1 + MD_TRACE_N); i += 256) {
                const int w_ = i / (1 + MD_TRACE_N), k_ = i - w_ * (1 + MD_TRACE_N);
                D.trace[i] = k_ == 0 ? (unsigned long long)g_md_trace_n[w_] : g_md_trace[w_][k_ - 1];
            }
        }
        __syncthreads();
#endif
        if (threadIdx.x == 0) { /* the LCU's neighbour state is in the maps: the next LCUs' mode decisions may start */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(&md_done[lcu], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (PROF && D.prof && threadIdx.x == 0) {
            c_md = __builtin_readcyclecounter();
            unsigned long long *q = D.prof + 16 * (size_t)lcu;
            for (int k = 0; k < 9; k++)
                q[k] += M.prof[k];
            q[9] += c_md - M.prof_t;         /* the LCU's state leaving LDS */
            q[12] += c_wait - c_ticket;       /* waiting for the LCU's neighbours */
            q[13] += M.prof[13], q[14] += M.prof[14]; /* candidates of the fast loops / units tested */
            unsigned long long *q2 = D.prof + 16 * (size_t)D.prof_lcus + 16 * (size_t)lcu; /* the sub-stage sums follow the stage sums of all LCUs */
            for (int k = 0; k < 16; k++)
                q2[k] += M.prof[16 + k];
            unsigned long long *q3 = D.prof + 32 * (size_t)D.prof_lcus + 128 * (size_t)lcu; /* ... and both by depth */
            for (int k = 0; k < 128; k++)
                q3[k] += M.prof_d[0][k];
            q[15] += 1;
        }
        if (D.encode) {
            if constexpr (INTER)
                md_ep_kinds(D, P, M, lx * 64, ly * 64);
            md_make_work<INTER, T>(D, P, M, lx * 64, ly * 64, works[lcu]);
            if (PROF && D.prof && threadIdx.x == 0)
                D.prof[16 * (size_t)lcu + 10] += __builtin_readcyclecounter() - c_md; /* merge / skip decisions with chroma + the work record */
        }
    }
}

static_assert(sizeof(MdShared<true>) + sizeof(MdPictureDev) + 64 <= 160 * 1024 && sizeof(MdShared<false>) + sizeof(MdPictureDev) + 64 <= 160 * 1024,
              "the LCU state has to fit the 160 KB of LDS of a CU");

/* ---- host side ------------------------------------------------------------------------------------------------------------- */
extern "C" int svt_amd_md_picture_supported(const SvtAmdMdPicture *P) { return P ? md_picture_supported(P) : 0; }
extern "C" int svt_amd_md_picture_supported_inter(const SvtAmdMdPicture *P, const SvtAmdMdInter *X) { return P && X ? md_picture_supported_inter(P, X) : 0; }
extern "C" int svt_amd_md_lcus_supported(const SvtAmdMdPicture *P, const SvtAmdMdLcu *lcus, int n)
{
    if (!P || !lcus)
        return 0;
    for (int i = 0; i < n; i++)
        if (!md_lcu_supported(P, &lcus[i]))
            return 0;
    return 1;
}

/* the mode decision's part of a picture object: neighbour maps, the picture's source and the per-LCU arrays, all in HBM */
struct SvtAmdMdState {
    MdPictureDev d;
    uint8_t *d_src[3];
    size_t src_cap[3];             /* bytes of d_src[k] (grown when a caller's row pitch needs more than the picture object's own) */
    SvtAmdOisLcuResult *d_ois;
    SvtAmdMdLcu *d_lcus;
    SvtAmdMdPicture *d_P;
    SvtAmdMdLcuOut *d_out;
    void *d_works, *d_results;    /* SvtAmdLcuWork / SvtAmdLcuResult of the picture object's sample width */
    size_t work_bytes, result_bytes;
    uint16_t *d_src16[3];          /* 10-bit pictures: the source as the encode pass codes it (d_src = its 8 MSBs, what the mode decision reads) */
    uint8_t *d_ref8[2][3];         /* ... and the 8-MSB views of the two reference pictures (UnPackReferenceBlock) */
    size_t ref8_bytes[2][3];
    size_t info_bytes, mv_bytes;
    unsigned long long *d_prof;
    unsigned long long *d_trace;
    int trace_lcu, trace_unit;
    int force_butterflies;
    unsigned *d_md_done;           /* epoch of the call whose mode decision finished the LCU */
    unsigned *d_md_ticket;         /* the mode-decision kernel's ticket counter (the encode pass behind it draws from the picture object's own) */
    MdPictureDev *d_D;             /* the descriptor the kernel reads (d, copied per call) */
    hipEvent_t ev_k0, ev_k1, ev_k2; /* around the last launch of k_md_picture on the call's stream (svt_amd_debug_md_kernel_ms), and behind the encode pass that follows it */
    int grid;                      /* its workgroups */
    /* P / B pictures */
    SvtAmdMdInter *d_X;
    SvtAmdMeLcuResult *d_me;
    SvtAmdTmvpLcu *d_tmvp;
    /* the call's small inputs (picture controls, inter controls, rate tables) come from the caller's stack / the encoder's heap: pageable memory, whose "asynchronous"
     * copy is a staged one the runtime completes inside the call - behind whatever its queue is running (profiles/r05_v: plane copies of one call issued 720 ms apart
     * while other pictures' mode-decision kernels ran).  They go through this page-locked block instead */
    uint8_t *h_stage;
    void *retired[8];              /* source planes outgrown by a caller's row pitch */
    int n_retired;
};
static constexpr size_t MD_STAGE_P = 0, MD_STAGE_X = (sizeof(SvtAmdMdPicture) + 63) & ~(size_t)63, MD_STAGE_COST = MD_STAGE_X + ((sizeof(SvtAmdMdInter) + 63) & ~(size_t)63),
                        MD_STAGE_D = MD_STAGE_COST + ((sizeof(SvtAmdCabacCost) + 63) & ~(size_t)63), MD_STAGE_BYTES = MD_STAGE_D + ((sizeof(MdPictureDev) + 63) & ~(size_t)63);

void svt_amd_md_state_free(SvtAmdEncDecPicture *pic)
{
    SvtAmdMdState *m = pic->md;
    if (!m)
        return;
    void *ptrs[] = {m->d.md_rec, m->d.md_info, m->d_src[0], m->d_src[1], m->d_src[2], m->d_ois, m->d_lcus, m->d_P, m->d_out, m->d_works, m->d_results,
                    m->d.md_mv, m->d_X, m->d_me, m->d_tmvp, m->d_prof, m->d_trace, m->d_md_done, m->d_md_ticket, m->d_D, m->d_src16[0], m->d_src16[1], m->d_src16[2],
                    m->d_ref8[0][0], m->d_ref8[0][1], m->d_ref8[0][2], m->d_ref8[1][0], m->d_ref8[1][1], m->d_ref8[1][2]};
    for (void *q : ptrs)
        if (q)
            (void)hipFree(q);
    for (int i = 0; i < m->n_retired; i++)
        (void)hipFree(m->retired[i]);
    if (m->ev_k0)
        (void)hipEventDestroy(m->ev_k0);
    if (m->ev_k1)
        (void)hipEventDestroy(m->ev_k1);
    if (m->ev_k2)
        (void)hipEventDestroy(m->ev_k2);
    if (m->h_stage)
        (void)hipHostFree(m->h_stage);
    free(m);
    pic->md = nullptr;
}

static int md_state(SvtAmdEncDecPicture *pic, SvtAmdMdState **out)
{
    if (pic->md) {
        *out = pic->md;
        return SVT_AMD_OK;
    }
    SvtAmdMdState *m = (SvtAmdMdState *)calloc(1, sizeof(*m));
    if (!m)
        return SVT_AMD_ERR_RESOURCES;
    pic->md = m;
    const size_t n = (size_t)pic->nlcu;
    m->d.md_pitch = pic->d.pitch[0];
    m->d.info_pitch = ((uint32_t)(pic->d.width >> 2) + 63) & ~63u;
    m->info_bytes = sizeof(uint32_t) * (size_t)m->d.info_pitch * (pic->d.height >> 2);
    const size_t bps = pic->d.bps;
    m->work_bytes = bps == 2 ? sizeof(SvtAmdLcuWork16) : sizeof(SvtAmdLcuWork), m->result_bytes = bps == 2 ? sizeof(SvtAmdLcuResult16) : sizeof(SvtAmdLcuResult);
    bool ok = hipMalloc((void **)&m->d.md_rec, pic->plane_bytes[0] / bps) == hipSuccess && hipMalloc((void **)&m->d.md_info, m->info_bytes) == hipSuccess;
    for (int k = 0; k < 3 && ok; k++) {
        /* room for the caller's row pitch (an encoder's padded input pictures: width + up to 512 columns): growing the plane later means a hipFree, which waits for every
         * kernel on the device - the other pictures' mode decisions - with the runtime's memory lock held (profiles/r05_aa: 90 - 300 ms, other threads' copies queueing behind it) */
        const size_t roomy = (size_t)((pic->d.width + 512) >> (k ? 1 : 0)) * (pic->d.height >> (k ? 1 : 0)) + 64;
        const size_t cap = roomy > pic->plane_bytes[k] / bps ? roomy : pic->plane_bytes[k] / bps;
        ok = hipMalloc((void **)&m->d_src[k], cap) == hipSuccess;
        m->src_cap[k] = ok ? cap - 64 : 0;
        if (bps == 2)
            ok = ok && hipMalloc((void **)&m->d_src16[k], pic->plane_bytes[k]) == hipSuccess;
    }
    ok = ok && hipMalloc((void **)&m->d_ois, sizeof(SvtAmdOisLcuResult) * n) == hipSuccess && hipMalloc((void **)&m->d_lcus, sizeof(SvtAmdMdLcu) * n) == hipSuccess &&
         hipMalloc((void **)&m->d_P, sizeof(SvtAmdMdPicture)) == hipSuccess && hipMalloc((void **)&m->d_out, sizeof(SvtAmdMdLcuOut) * n) == hipSuccess &&
         hipMalloc((void **)&m->d_works, m->work_bytes * n) == hipSuccess && hipMalloc((void **)&m->d_results, m->result_bytes * n) == hipSuccess;
    m->d.mv_pitch = ((uint32_t)(pic->d.width >> 3) + 15) & ~15u;
    m->mv_bytes = sizeof(uint4) * (size_t)m->d.mv_pitch * ((pic->d.height + 7) >> 3);
    ok = ok && hipMalloc((void **)&m->d.md_mv, m->mv_bytes) == hipSuccess && hipMalloc((void **)&m->d_X, sizeof(SvtAmdMdInter)) == hipSuccess &&
         hipMalloc((void **)&m->d_me, sizeof(SvtAmdMeLcuResult) * n) == hipSuccess && hipMalloc((void **)&m->d_tmvp, sizeof(SvtAmdTmvpLcu) * (n + 1)) == hipSuccess;
    ok = ok && hipMalloc((void **)&m->d_md_done, sizeof(unsigned) * n) == hipSuccess && hipMemset(m->d_md_done, 0, sizeof(unsigned) * n) == hipSuccess;
    ok = ok && hipMalloc((void **)&m->d_md_ticket, 64) == hipSuccess && hipMalloc((void **)&m->d_D, sizeof(MdPictureDev)) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&m->h_stage, MD_STAGE_BYTES, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipEventCreate(&m->ev_k0) == hipSuccess && hipEventCreateWithFlags(&m->ev_k1, hipEventBlockingSync) == hipSuccess; /* ev_k1: the caller sleeps through the kernel */
    ok = ok && hipEventCreate(&m->ev_k2) == hipSuccess;
    if (!ok) {
        svt_amd_set_error("hipMalloc (mode-decision picture state) failed");
        svt_amd_md_state_free(pic);
        return SVT_AMD_ERR_RESOURCES;
    }
    *out = m;
    return SVT_AMD_OK;
}

/* the 8 most significant bits of 10-bit samples in 16-bit words (UnPack8BitDataSafeSub, C_DEFAULT/EbPackUnPack_C.c:203: sample >> 2): eight samples a thread */
__global__ __launch_bounds__(256) void k_msb_view(const uint16_t *__restrict__ in, uint8_t *__restrict__ out, size_t n)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const uint4 v = *(const uint4 *)(in + i);
        uint2 o;
        o.x = ((v.x >> 2) & 0xffu) | ((v.x >> 10) & 0xff00u) | (((v.y >> 2) & 0xffu) << 16) | (((v.y >> 18) & 0xffu) << 24);
        o.y = ((v.z >> 2) & 0xffu) | ((v.z >> 10) & 0xff00u) | (((v.w >> 2) & 0xffu) << 16) | (((v.w >> 18) & 0xffu) << 24);
        *(uint2 *)(out + i) = o;
    } else {
        for (size_t k = i; k < n; k++)
            out[k] = (uint8_t)(in[k] >> 2);
    }
}
/* a source plane from page-locked HOST memory into HBM by a kernel of the call's own stream (the device reads the host over PCIe): the runtime's
 * hipMemcpyAsync of such a plane - a DMA-engine transfer - blocked the calling thread for one to four mode-decision kernel durations whenever other
 * pictures' kernels were running (profiles/r05_y: 50 - 420 ms inside the call, a tenth of the calls); a launch never does.  src: any alignment; dst: 16 bytes */
typedef uint4 __attribute__((aligned(1))) md_u128u;
__global__ __launch_bounds__(256) void k_md_fetch_plane(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, size_t bytes)
{
    const size_t n16 = bytes >> 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        *(uint4 *)(dst + i * 16) = *(const md_u128u *)(src + i * 16);
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15))
        dst[(n16 << 4) + threadIdx.x] = src[(n16 << 4) + threadIdx.x];
}
static int msb_view(hipStream_t st, const void *in16, uint8_t *out8, size_t samples)
{
    if (!samples)
        return SVT_AMD_OK;
    hipLaunchKernelGGL(k_msb_view, dim3((unsigned)((samples + 2047) / 2048)), dim3(256), 0, st, (const uint16_t *)in16, out8, samples);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}

/* How much of the device the k_md_encode_picture launches in flight may hold.  A workgroup of this kernel holds a whole CU (its LDS) for the life of the launch, most of it
 * waiting for neighbours: whatever else the encoder needs meanwhile - the next pictures' motion estimation and open-loop intra search, picture preparation, the runtime's
 * copy and fill kernels - waits for a mode-decision launch to END once they fill the part (profiles/r05_a: with 12 - 16 pictures in the encoder's pool the closed loop
 * fell from 57 to 25 fps, launches starting in bursts exactly when others finished).  The launches therefore share a BUDGET of workgroups = CUs - CUs / 16 (240 on an
 * MI355X): a call waits for its grid to fit AFTER its inputs are on the device (lowest temporal layer first among the waiters, then first come) and gives the workgroups
 * back when its kernel has finished, before its records travel back.
 * How wide a launch is decides how many pictures share the device: the wavefront of a 4K picture is 30 LCUs wide at its widest and 16 on average; 40 workgroups (the
 * widest front + the encode passes behind it) give a lone picture its shortest kernel (51.5 ms for a layer-2 picture), 20 cost it 9 % (56.0 ms) and let twelve pictures
 * run side by side - 148 instead of 96 pictures/s of device capacity (profiles/r05_k_md_flights_24q.txt), 93 - 96 instead of 79 fps of encode (r05_k_sweep_grid2.txt).
 * A call therefore takes the wide grid only while the device is nearly idle (at most a third of the budget in use and nobody waiting) and the narrow one (half) otherwise.
 * SVT_AMD_MD_MAX_KERNELS=<n> forces a plain count limit instead, SVT_AMD_MD_GRID=<n> a fixed width (measurement). */
namespace {
std::mutex g_flight_mu;
std::condition_variable g_flight_cv;
/* per DEVICE: a process with contexts on several GPUs must not charge device B's launches against device A's compute units */
struct FlightDevice { int flights = 0, wgs = 0, cus = 0; };
FlightDevice g_flight_dev[64];
unsigned long long g_flight_ticket;
struct FlightWaiter { int prio; unsigned long long ticket; int device; };
std::vector<FlightWaiter> g_flight_wait;
int md_forced_count_limit()
{
    static const int forced = getenv("SVT_AMD_MD_MAX_KERNELS") ? atoi(getenv("SVT_AMD_MD_MAX_KERNELS")) : 0;
    return forced;
}
int md_wg_budget(int device)
{
    int c;
    {
        std::lock_guard<std::mutex> l(g_flight_mu);
        FlightDevice &fd = g_flight_dev[device & 63];
        if (!fd.cus) {
            hipDeviceProp_t pr;
            fd.cus = hipGetDeviceProperties(&pr, device) == hipSuccess ? pr.multiProcessorCount : 256;
        }
        c = fd.cus;
    }
    static const int forced = getenv("SVT_AMD_MD_WG_BUDGET") ? atoi(getenv("SVT_AMD_MD_WG_BUDGET")) : 0; /* measurement */
    return forced > 0 ? forced : c - c / 16;
}
/* SVT_AMD_MD_FLIGHT_STATS=<file> (measurement): one line appended when the process ends - calls, time spent waiting for a place, wide / narrow grants */
struct FlightStats {
    std::atomic<unsigned long long> calls{0}, wide{0}, wait_us{0}, run_us{0}, wgs_seen{0};
    ~FlightStats()
    {
        const char *path = getenv("SVT_AMD_MD_FLIGHT_STATS");
        FILE *f = calls && path ? fopen(path, "a") : nullptr;
        if (!f)
            return;
        fprintf(f, "svt_amd: mode-decision launches: %llu calls (%llu at the wide grid), waiting for a place %.1f ms a call, launch -> kernel done %.1f ms a call, "
                   "%.1f workgroups of other calls in flight at a grant (mean)\n", (unsigned long long)calls, (unsigned long long)wide, 1e-3 * (double)wait_us / (double)calls,
                1e-3 * (double)run_us / (double)calls, (double)wgs_seen / (double)calls);
        fclose(f);
    }
} g_flight_stats;
double flight_now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct MdFlight {
    int held = 0; /* workgroups this call holds */
    int dev = 0;
    double t_granted = 0;
    /* waits until the launch fits; returns the grid granted: `wide` when the device is nearly idle, `narrow` otherwise */
    /* lone: the width of a launch that finds the device without any other mode-decision launch (two workgroups per LCU of the widest front); wide / narrow: round 5's pair for
     * a device that is shared - the encoder's steady state is untouched by `lone` */
    int acquire(int device, int budget, int lone, int wide, int narrow, int prio)
    {
        const double t0 = flight_now_us();
        dev = device & 63;
        std::unique_lock<std::mutex> l(g_flight_mu);
        FlightDevice &fd = g_flight_dev[dev];
        const FlightWaiter me = {prio, g_flight_ticket++, dev};
        g_flight_wait.push_back(me);
        const int count_limit = md_forced_count_limit();
        int grant = 0;
        g_flight_cv.wait(l, [&] {
            int waiting_here = 0;
            for (const FlightWaiter &w : g_flight_wait) {
                if (w.device != dev)
                    continue;
                waiting_here++;
                if (w.prio < me.prio || (w.prio == me.prio && w.ticket < me.ticket))
                    return false;
            }
            if (count_limit > 0) {
                grant = wide;
                return fd.flights < count_limit;
            }
            grant = (waiting_here == 1 && fd.flights == 0) ? lone : (waiting_here == 1 && fd.wgs + wide <= budget / 3) ? wide : narrow;
            return fd.wgs + grant <= budget || fd.flights == 0;
        });
        for (size_t i = 0; i < g_flight_wait.size(); i++)
            if (g_flight_wait[i].ticket == me.ticket) {
                g_flight_wait.erase(g_flight_wait.begin() + (long)i);
                break;
            }
        g_flight_stats.calls++, g_flight_stats.wide += (grant == wide || grant == lone) && wide != narrow, g_flight_stats.wgs_seen += (unsigned long long)fd.wgs;
        fd.flights++, fd.wgs += grant, held = grant;
        l.unlock();
        t_granted = flight_now_us();
        g_flight_stats.wait_us += (unsigned long long)(t_granted - t0);
        g_flight_cv.notify_all(); /* the next in line may fit as well */
        return grant;
    }
    void release()
    {
        if (!held)
            return;
        g_flight_stats.run_us += (unsigned long long)(flight_now_us() - t_granted);
        {
            std::lock_guard<std::mutex> l(g_flight_mu);
            g_flight_dev[dev].flights--, g_flight_dev[dev].wgs -= held;
        }
        held = 0;
        g_flight_cv.notify_all();
    }
    ~MdFlight() { release(); }
};
}

/* the kernel's dynamic LDS size, once per device */
static int md_kernel_attributes(int device)
{
    static std::mutex mu;
    static bool attr[64];
    std::lock_guard<std::mutex> g(mu);
    if (!attr[device & 63]) {
        /* (the kernels' LCU state is static shared memory: nothing to raise) */
        attr[device & 63] = true;
    }
    return SVT_AMD_OK;
}
/* everything the first svt_amd_md_encode_picture[_inter] call on this picture object would allocate (device state, page-locked staging, events), made now - by a host that
 * builds its picture objects before the clock starts (the encoder binding: EbInitEncoder) */
extern "C" int svt_amd_md_picture_warmup(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic)
{
    if (!ctx || !pic)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rate_tables_once(ctx->device);
    SvtAmdMdState *m = nullptr;
    if (!rc)
        rc = md_state(pic, &m);
    if (!rc)
        rc = md_kernel_attributes(ctx->device);
    return rc;
}

/* bps: bytes per sample of the source planes, the work / result records and the picture object (1, or 2 = a 10-bit picture: the mode decision on the 8 MSBs of source
 * and reference pictures, the encode pass on the 10-bit samples) */
static int md_encode_picture(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdMdPicture *P, const SvtAmdMdInter *X, const SvtAmdMdLcu *lcus,
                             const void *src_y, uint32_t stride_y, const void *src_cb, const void *src_cr, uint32_t stride_c, const SvtAmdOisLcuResult *ois,
                             int ois_slot, const SvtAmdMeLcuResult *me, int me_slot, const SvtAmdTmvpLcu *tmvp, const SvtAmdCabacCost *cost, SvtAmdMdLcuOut *md_out,
                             void *works, void *results, uint32_t bps)
{
    if (!ctx || !pic || !P || !lcus || !src_y || !src_cb || !src_cr || (!X && !cost))
        return SVT_AMD_ERR_BAD_PARAM;
    if ((X ? !md_picture_supported_inter(P, X) : !md_picture_supported(P)) || P->width != pic->d.width || P->height != pic->d.height || pic->d.bps != bps) {
        svt_amd_set_error("svt_amd_md_encode_picture: picture outside what this revision covers (svt_amd_md_picture_supported[_inter]), or not the picture object's size / sample width");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    if (X && (!pic->has_cost || !pic->has_ref[0] || (P->slice_type == 0 && !pic->has_ref[1]) || (X->tmvp_enable && !tmvp))) {
        svt_amd_set_error("svt_amd_md_encode_picture_inter: reference pictures / rate tables not set (svt_amd_encdec_picture_set_inter), or no co-located motion field");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    const int n = pic->nlcu, wl = (pic->d.width + 63) / 64, hl = (pic->d.height + 63) / 64;
    int tiles = 0;
    for (int i = 0; i < n; i++) {
        if (lcus[i].leaf_count < 1 || lcus[i].leaf_count > SVT_AMD_MD_LEAVES) {
            svt_amd_set_error("svt_amd_md_encode_picture: LCU %d has no leaf list", i);
            return SVT_AMD_ERR_BAD_PARAM;
        }
        for (int k = 0; k < lcus[i].leaf_count; k++)
            if (lcus[i].leaf_index[k] >= SVT_AMD_MD_LEAVES || (k && lcus[i].leaf_index[k] <= lcus[i].leaf_index[k - 1])) {
                svt_amd_set_error("svt_amd_md_encode_picture: LCU %d: leaf list not ascending", i);
                return SVT_AMD_ERR_BAD_PARAM;
            }
        tiles += lcus[i].tile_left && lcus[i].tile_top;
        if (X && !md_lcu_supported(P, &lcus[i])) {
            svt_amd_set_error("svt_amd_md_encode_picture_inter: LCU %d is not decided by ModeDecisionLcu with luma-only candidates", i);
            return SVT_AMD_ERR_BAD_PARAM;
        }
    }
    hipEvent_t ev_wait[2] = {nullptr, nullptr}; /* records read where another lane's kernels leave them: this lane's stream orders itself behind those kernels */
    const SvtAmdMeLcuResult *d_me_slot = nullptr;
    if (X && !me) {
        SvtAmdContext *root = ctx->parent ? ctx->parent : ctx;
        if (me_slot < 0 || me_slot >= root->num_slots || !root->slots[me_slot].d_me_out || !root->slots[me_slot].valid || __atomic_load_n(&root->slots[me_slot].me_lcus, __ATOMIC_ACQUIRE) != (uint32_t)n ||
            root->slots[me_slot].width != pic->d.width || root->slots[me_slot].height != pic->d.height) {
            svt_amd_set_error("svt_amd_md_encode_picture_inter: slot %d does not hold the motion-estimation records of a %u x %u picture (none launched for the slot's current picture, or only a part of it)", me_slot, pic->d.width, pic->d.height);
            return SVT_AMD_ERR_BAD_PARAM;
        }
        d_me_slot = root->slots[me_slot].d_me_out;
        ev_wait[0] = root->slots[me_slot].ev_me;
    }
    const SvtAmdOisLcuResult *d_ois_slot = nullptr;
    if (!ois) {
        SvtAmdContext *root = ctx->parent ? ctx->parent : ctx;
        if (ois_slot < 0 || ois_slot >= root->num_slots || !root->slots[ois_slot].d_ois_out || !root->slots[ois_slot].valid || __atomic_load_n(&root->slots[ois_slot].ois_lcus, __ATOMIC_ACQUIRE) != (uint32_t)n ||
            root->slots[ois_slot].width != pic->d.width || root->slots[ois_slot].height != pic->d.height) {
            svt_amd_set_error("svt_amd_md_encode_picture: slot %d does not hold the open-loop intra search records of a %u x %u picture", ois_slot, pic->d.width, pic->d.height);
            return SVT_AMD_ERR_BAD_PARAM;
        }
        d_ois_slot = root->slots[ois_slot].d_ois_out;
        ev_wait[1] = root->slots[ois_slot].ev_ois;
    }
    if (pic->md_rect_n) { /* a rank's rectangle (svt_amd_encdec_picture_set_rect): its borders must be tile borders - an LCU never waits for one outside */
        const SvtAmdRect &r = pic->md_rect;
        const int x0 = r.x / 64, y0 = r.y / 64, x1 = (r.x + r.w + 63) / 64, y1 = (r.y + r.h + 63) / 64;
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                const SvtAmdMdLcu &Lc = lcus[y * wl + x];
                if ((x == x0 && !Lc.tile_left) || (y == y0 && !Lc.tile_top) || (x == x1 - 1 && x1 < wl && !Lc.tile_right) ||
                    (y == y1 - 1 && y1 < hl && !lcus[(y + 1) * wl + x].tile_top)) {
                    svt_amd_set_error("svt_amd_md_encode_picture: the rectangle's border at LCU (%d, %d) is not a tile border", x, y);
                    return SVT_AMD_ERR_BAD_PARAM;
                }
            }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rate_tables_once(ctx->device);
    if (rc)
        return rc;
    SvtAmdMdState *m = nullptr;
    if ((rc = md_state(pic, &m)) != 0)
        return rc;
    hipStream_t st = ctx->stream;
    for (hipEvent_t ev : ev_wait)
        if (ev)
            HIP_TRY(hipStreamWaitEvent(st, ev, 0));
    /* debug (SVT_AMD_MD_TIMING): host clock around the call's three parts, with a stream synchronisation after each - one line per call on stderr */
    static const bool timing = getenv("SVT_AMD_MD_TIMING") != nullptr;
    const auto t_call = std::chrono::steady_clock::now();
    if (timing)
        HIP_TRY(hipStreamSynchronize(st)); /* what the stream still had to do + the front half's records */
    const auto t_begin = std::chrono::steady_clock::now();
    auto t_up = t_begin, t_kernel = t_begin;
    auto t_tick = t_begin;
    auto tick = [&](const char *what) { /* SVT_AMD_MD_TIMING: a host call of the upload section that took more than 5 ms names itself */
        if (!timing)
            return;
        const auto now = std::chrono::steady_clock::now();
        const double d = std::chrono::duration<double, std::milli>(now - t_tick).count();
        if (d > 5.0)
            fprintf(stderr, "svt_amd_md_encode_picture: %.1f ms inside the host call '%s'\n", d, what);
        t_tick = now;
    };
    const void *hs[3] = {src_y, src_cb, src_cr};
    /* 8-bit source planes travel as ONE linear transfer each, in the host's row pitch (the device copy keeps that pitch): a strided copy is a shader kernel of the runtime
     * (__amd_rocclr_copyBufferRect: 2 ms per plane, and 40 ms when the mode-decision launches of other pictures hold the CUs - profiles/r05_b), a linear one from page-locked
     * memory is the DMA engine's */
    const bool linear = bps == 1 && stride_y % 4 == 0 && stride_c % 4 == 0 && stride_y < 2 * pic->d.width + 256 && stride_c < pic->d.width + 256;
    for (int k = 0; k < 3; k++) {
        const uint32_t pw = k ? pic->d.width / 2 : pic->d.width, ph = k ? pic->d.height / 2 : pic->d.height;
        if (linear) {
            const size_t hstride = k ? stride_c : stride_y, need = hstride * (ph - 1) + pw;
            if (m->src_cap[k] < need) {
                HIP_TRY(hipStreamSynchronize(st));
                if (m->d_src[k] && m->n_retired < 8)
                    m->retired[m->n_retired++] = m->d_src[k]; /* freed with the state: a hipFree here would wait for every kernel on the device */
                else if (m->d_src[k])
                    (void)hipFree(m->d_src[k]);
                m->d_src[k] = nullptr, m->src_cap[k] = 0;
                HIP_TRY(hipMalloc((void **)&m->d_src[k], need + 64));
                m->src_cap[k] = need;
            }
            const uint8_t *dv = (const uint8_t *)svt_amd_registered_device_ptr(hs[k], need);
            if (dv) {
                hipLaunchKernelGGL(k_md_fetch_plane, dim3(k ? 32 : 96), dim3(256), 0, st, m->d_src[k], dv, need);
                HIP_TRY(hipGetLastError());
            } else {
                HIP_TRY(hipMemcpyAsync(m->d_src[k], hs[k], need, hipMemcpyHostToDevice, st));
            }
            tick(k ? "source chroma plane copy" : "source luma plane copy");
        } else if (bps == 1) {
            HIP_TRY(hipMemcpy2DAsync(m->d_src[k], pic->d.pitch[k], hs[k], k ? stride_c : stride_y, pw, ph, hipMemcpyHostToDevice, st));
        } else { /* strides in samples */
            HIP_TRY(hipMemcpy2DAsync(m->d_src16[k], (size_t)pic->d.pitch[k] * 2, hs[k], (size_t)(k ? stride_c : stride_y) * 2, (size_t)pw * 2, ph, hipMemcpyHostToDevice, st));
            if ((rc = msb_view(st, m->d_src16[k], m->d_src[k], pic->plane_bytes[k] / 2)) != 0)
                return rc;
        }
        m->d.src[k] = m->d_src[k], m->d.src16[k] = bps == 2 ? m->d_src16[k] : nullptr;
    }
    m->d.src_pitch[0] = linear ? stride_y : pic->d.pitch[0], m->d.src_pitch[1] = linear ? stride_c : pic->d.pitch[1];
    for (int l = 0; l < 2; l++) { /* the reference pictures as the mode decision reads them */
        m->d.mref[l] = pic->d.ref[l];
        if (bps == 2 && X && pic->has_ref[l])
            for (int p = 0; p < 3; p++) {
                const size_t need = (size_t)pic->d.ref[l].size[p ? 1 : 0];
                if (m->ref8_bytes[l][p] < need) {
                    if (m->d_ref8[l][p])
                        (void)hipFree(m->d_ref8[l][p]);
                    m->d_ref8[l][p] = nullptr, m->ref8_bytes[l][p] = 0;
                    HIP_TRY(hipMalloc((void **)&m->d_ref8[l][p], need + 16));
                    m->ref8_bytes[l][p] = need;
                }
                if ((rc = msb_view(st, pic->d.ref[l].plane[p], m->d_ref8[l][p], need)) != 0)
                    return rc;
                m->d.mref[l].plane[p] = m->d_ref8[l][p];
            }
    }
    HIP_TRY(hipMemcpyAsync(m->d_lcus, lcus, sizeof(SvtAmdMdLcu) * (size_t)n, hipMemcpyHostToDevice, st));
    tick("LCU controls copy");
    memcpy(m->h_stage + MD_STAGE_P, P, sizeof(*P)); /* (the block is free: the previous call on this object returned after its stream had drained) */
    HIP_TRY(hipMemcpyAsync(m->d_P, m->h_stage + MD_STAGE_P, sizeof(*P), hipMemcpyHostToDevice, st));
    if (cost) {
        memcpy(m->h_stage + MD_STAGE_COST, cost, sizeof(*cost));
        HIP_TRY(hipMemcpyAsync(pic->d_cost, m->h_stage + MD_STAGE_COST, sizeof(*cost), hipMemcpyHostToDevice, st));
        pic->has_cost = true;
    }
    m->d.X = nullptr, m->d.me = nullptr, m->d.tmvp = nullptr;
    if (X) {
        memcpy(m->h_stage + MD_STAGE_X, X, sizeof(*X));
        HIP_TRY(hipMemcpyAsync(m->d_X, m->h_stage + MD_STAGE_X, sizeof(*X), hipMemcpyHostToDevice, st));
        tick("picture / inter controls copies");
        if (me)
            HIP_TRY(hipMemcpyAsync(m->d_me, me, sizeof(SvtAmdMeLcuResult) * (size_t)n, hipMemcpyHostToDevice, st));
        tick("ME records copy");
        if (X->tmvp_enable) {
            HIP_TRY(hipMemcpyAsync(m->d_tmvp, tmvp, sizeof(SvtAmdTmvpLcu) * (size_t)n, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemsetAsync(m->d_tmvp + n, 0, sizeof(SvtAmdTmvpLcu), st));
            tick("motion-field copy + fill");
        }
        m->d.X = m->d_X, m->d.me = me ? m->d_me : d_me_slot, m->d.tmvp = m->d_tmvp;
        HIP_TRY(hipMemsetAsync(m->d.md_mv, 0, m->mv_bytes, st));
        tick("mv fill");
    }
    int n_active = n;
    const unsigned *d_order = pic->d_sync + 1 + n;
    if (pic->md_rect_n) {
        n_active = pic->md_rect_n, d_order = pic->d_order_md;
        HIP_TRY(hipMemsetAsync(m->d_out, 0, sizeof(SvtAmdMdLcuOut) * (size_t)n, st));
    }
    m->d.prof = m->d_prof, m->d.prof_lcus = n;
    m->d.trace = m->d_trace, m->d.trace_lcu = m->trace_lcu, m->d.trace_unit = m->trace_unit;
    m->d.force_butterflies = m->force_butterflies;
    m->d.encode = !X || works || results; /* P / B pictures: without a place for the work / result records, the mode decision alone */
    if (ois)
        HIP_TRY(hipMemcpyAsync(m->d_ois, ois, sizeof(SvtAmdOisLcuResult) * (size_t)n, hipMemcpyHostToDevice, st));
    m->d.ois = ois ? m->d_ois : d_ois_slot, m->d.lcus = m->d_lcus, m->d.P = m->d_P, m->d.out = m->d_out;
    pic->epoch++;
    pic->deblocked = pic->sao_done = false;
    HIP_TRY(hipMemsetAsync(pic->d_sync, 0, sizeof(unsigned), st));
    HIP_TRY(hipMemsetAsync(pic->d.mode_map, 0xFF, pic->map_bytes, st));
    HIP_TRY(hipMemsetAsync(m->d.md_info, 0xFF, m->info_bytes, st));
    HIP_TRY(hipMemsetAsync(m->d_results, 0, m->result_bytes * (size_t)n, st));
    HIP_TRY(hipMemsetAsync(m->d_works, 0, m->work_bytes * (size_t)n, st));
    tick("OIS copy + five fills");
    if ((rc = md_kernel_attributes(ctx->device)) != 0)
        return rc;
    if (timing) {
        HIP_TRY(hipStreamSynchronize(st));
        tick("stream synchronisation after the uploads");
        t_up = std::chrono::steady_clock::now();
    }
    /* the wavefront is at most min((W/64 + 1) / 2, H/64) LCUs wide; the mode decision runs ahead of the encode pass, so twice that many workgroups
     * find work (all of them resident: a workgroup that waits holds its CU) */
    /* ... but a workgroup holds a whole CU (its LDS) while it waits, and the encoder keeps several pictures in flight: five 62-workgroup launches do not fit the 256 CUs.
     * The wavefront of a picture is (W/64 + 1) / 2 LCUs wide at its widest and 16 on average at 4K; one workgroup per LCU of the widest front plus a few for the encode
     * passes behind it costs a lone picture 3 - 4 % and lets six pictures' launches run side by side (profiles/r04_w_md_flights_*: the kernel keeps its 51 ms with
     * twelve calls in flight, 100 pictures/s - given enough hardware queues, svt_amd_runtime_env_defaults). */
    /* Round 6: a lone picture on an idle device gets two workgroups per LCU of the widest front again - behind an LCU's decisions its workgroup spends ~15 % of the LCU's
     * time on the merge / skip decisions with chroma and the work record, ahead of them on the next LCU's inputs, and with one workgroup per LCU of the front a ready LCU
     * waits for a workgroup (4K: 40 / 48 / 56 / 64 workgroups = 27.1 / 27.05 / 26.85 / 26.7 ms for a layer-2 picture).  The widths a launch gets while other pictures' launches
     * are on the device - the encoder's steady state - stay what round 5 measured best (40 / 20 at 4K). */
    const int front = ((wl + 1) / 2 < hl ? (wl + 1) / 2 : hl) * (tiles > 0 ? tiles : 1);
    int lone = 2 * front + 2;
    int grid = front + 2 + (wl + 7) / 8;
    int narrow = (grid + 1) / 2 < 8 ? 8 : (grid + 1) / 2; /* what the launch gets while other pictures share the device (MdFlight) */
    {   /* measurement (SVT_AMD_MD_GRID): a fixed launch width */
        const char *fg = getenv("SVT_AMD_MD_GRID");
        const int forced = fg ? atoi(fg) : 0;
        if (forced > 0)
            lone = grid = narrow = forced;
        static const int fnarrow = getenv("SVT_AMD_MD_NARROW") ? atoi(getenv("SVT_AMD_MD_NARROW")) : 0;
        if (fnarrow > 0)
            narrow = fnarrow;
        static const int fwide = getenv("SVT_AMD_MD_WIDE") ? atoi(getenv("SVT_AMD_MD_WIDE")) : 0;
        if (fwide > 0)
            lone = grid = fwide;
    }
    grid = grid > n_active ? n_active : grid > 224 ? 224 : grid;
    lone = lone > n_active ? n_active : lone > 224 ? 224 : lone;
    lone = lone < grid ? grid : lone;
    narrow = narrow > grid ? grid : narrow;
    m->d.cost = pic->has_cost ? pic->d_cost : nullptr;
    memcpy(m->h_stage + MD_STAGE_D, &m->d, sizeof(m->d));
    HIP_TRY(hipMemcpyAsync(m->d_D, m->h_stage + MD_STAGE_D, sizeof(m->d), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(m->d_md_ticket, 0, sizeof(unsigned), st));
    MdFlight flight;
    /* the inputs first (a call that waits for its place holds no CU and no copy engine meanwhile) */
    HIP_TRY(hipStreamSynchronize(st));
    grid = flight.acquire(ctx->device, md_wg_budget(ctx->device), lone, grid, narrow, (int)P->temporal_layer);
    m->grid = grid;
    HIP_TRY(hipEventRecord(m->ev_k0, st));
    const bool profiled = m->d.prof != nullptr; /* (the P / B kernels exist with and without the stage marks) */
    if (X && bps == 1 && !profiled)
        hipLaunchKernelGGL((k_md_picture<true, uint8_t, false>), dim3((unsigned)grid), dim3(256), 0, st, m->d_D, (SvtAmdLcuWork *)m->d_works, n_active, wl, m->d_md_ticket,
                           m->d_md_done, d_order, pic->epoch);
    else if (X && bps == 1)
        hipLaunchKernelGGL((k_md_picture<true, uint8_t, true>), dim3((unsigned)grid), dim3(256), 0, st, m->d_D, (SvtAmdLcuWork *)m->d_works, n_active, wl, m->d_md_ticket,
                           m->d_md_done, d_order, pic->epoch);
#ifdef MD_INTER8_ONLY
    else
        return SVT_AMD_ERR_BAD_PARAM;
#else
    else if (bps == 1)
        hipLaunchKernelGGL((k_md_picture<false, uint8_t, true>), dim3((unsigned)grid), dim3(256), 0, st, m->d_D, (SvtAmdLcuWork *)m->d_works, n_active, wl, m->d_md_ticket,
                           m->d_md_done, d_order, pic->epoch);
    else if (X && !profiled)
        hipLaunchKernelGGL((k_md_picture<true, uint16_t, false>), dim3((unsigned)grid), dim3(256), 0, st, m->d_D, (SvtAmdLcuWork16 *)m->d_works, n_active, wl, m->d_md_ticket,
                           m->d_md_done, d_order, pic->epoch);
    else if (X)
        hipLaunchKernelGGL((k_md_picture<true, uint16_t, true>), dim3((unsigned)grid), dim3(256), 0, st, m->d_D, (SvtAmdLcuWork16 *)m->d_works, n_active, wl, m->d_md_ticket,
                           m->d_md_done, d_order, pic->epoch);
    else
        hipLaunchKernelGGL((k_md_picture<false, uint16_t, true>), dim3((unsigned)grid), dim3(256), 0, st, m->d_D, (SvtAmdLcuWork16 *)m->d_works, n_active, wl, m->d_md_ticket,
                           m->d_md_done, d_order, pic->epoch);
#endif
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(m->ev_k1, st));
    /* the encode pass of the picture: its own kernel behind the decisions, on the same stream (every LCU's work record is in HBM; LCUs without intra units wait for nobody,
     * so a P / B picture runs as wide as the device is) */
    if (m->d.encode && (rc = svt_amd_ep_launch_behind_md(ctx, pic, m->d_works, m->d_results, n_active, d_order, X != nullptr, tiles)) != 0)
        return rc;
    HIP_TRY(hipEventRecord(m->ev_k2, st));
    if ((rc = ep_picture_written(ctx, pic)) != 0)
        return rc;
    {   /* records read in place: the next motion-estimation / open-loop intra launch INTO those slots orders itself behind this kernel (context.hip slot_records_before_write) */
        SvtAmdContext *root = ctx->parent ? ctx->parent : ctx;
        const int rs[2] = {d_me_slot ? me_slot : -1, d_ois_slot ? ois_slot : -1};
        for (int k = 0; k < 2; k++)
            if (rs[k] >= 0 && !(k == 1 && rs[1] == rs[0])) {
                HIP_TRY(hipEventRecord(root->slots[rs[k]].ev_md_read, st));
                __atomic_store_n(&root->slots[rs[k]].md_read_pending, 1, __ATOMIC_RELEASE);
            }
    }
    if (timing) {
        HIP_TRY(hipStreamSynchronize(st));
        t_kernel = std::chrono::steady_clock::now();
    }
    if (flight.held) { /* the workgroups are free as soon as the kernel has finished: the records' way back needs no CU */
        HIP_TRY(hipEventSynchronize(m->ev_k1));
        flight.release();
    }
    if (md_out)
        HIP_TRY(hipMemcpyAsync(md_out, m->d_out, sizeof(SvtAmdMdLcuOut) * (size_t)n, hipMemcpyDeviceToHost, st));
    if (works)
        HIP_TRY(hipMemcpyAsync(works, m->d_works, m->work_bytes * (size_t)n, hipMemcpyDeviceToHost, st));
    if (results)
        HIP_TRY(hipMemcpyAsync(results, m->d_results, m->result_bytes * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (timing) {
        const auto t_end = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "svt_amd_md_encode_picture%s: layer %d, %d LCUs, stream + front-half records ready %.2f ms, inputs up %.2f ms, kernel %.2f ms, records down %.2f ms\n", X ? "_inter" : "",
                (int)P->temporal_layer, n_active, ms(t_call, t_begin), ms(t_begin, t_up), ms(t_up, t_kernel), ms(t_kernel, t_end));
    }
    return SVT_AMD_OK;
}

extern "C" int svt_amd_md_encode_picture(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdMdPicture *P, const SvtAmdMdLcu *lcus, const uint8_t *src_y,
                                         uint32_t stride_y, const uint8_t *src_cb, const uint8_t *src_cr, uint32_t stride_c, const SvtAmdOisLcuResult *ois,
                                         int ois_slot, const SvtAmdCabacCost *cost, SvtAmdMdLcuOut *md_out, SvtAmdLcuWork *works, SvtAmdLcuResult *results)
{
    if (!cost)
        return SVT_AMD_ERR_BAD_PARAM;
    return md_encode_picture(ctx, pic, P, nullptr, lcus, src_y, stride_y, src_cb, src_cr, stride_c, ois, ois_slot, nullptr, -1, nullptr, cost, md_out, works, results, 1);
}
extern "C" int svt_amd_md_encode_picture16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdMdPicture *P, const SvtAmdMdLcu *lcus, const uint16_t *src_y,
                                           uint32_t stride_y, const uint16_t *src_cb, const uint16_t *src_cr, uint32_t stride_c, const SvtAmdOisLcuResult *ois,
                                           int ois_slot, const SvtAmdCabacCost *cost, SvtAmdMdLcuOut *md_out, SvtAmdLcuWork16 *works, SvtAmdLcuResult16 *results)
{
    if (!cost)
        return SVT_AMD_ERR_BAD_PARAM;
    return md_encode_picture(ctx, pic, P, nullptr, lcus, src_y, stride_y, src_cb, src_cr, stride_c, ois, ois_slot, nullptr, -1, nullptr, cost, md_out, works, results, 2);
}

extern "C" int svt_amd_md_encode_picture_inter(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdMdPicture *P, const SvtAmdMdInter *X, const SvtAmdMdLcu *lcus,
                                               const uint8_t *src_y, uint32_t stride_y, const uint8_t *src_cb, const uint8_t *src_cr, uint32_t stride_c,
                                               const SvtAmdOisLcuResult *ois, int ois_slot, const SvtAmdMeLcuResult *me, int me_slot, const SvtAmdTmvpLcu *tmvp,
                                               SvtAmdMdLcuOut *md_out, SvtAmdLcuWork *works, SvtAmdLcuResult *results)
{
    if (!X)
        return SVT_AMD_ERR_BAD_PARAM;
    return md_encode_picture(ctx, pic, P, X, lcus, src_y, stride_y, src_cb, src_cr, stride_c, ois, ois_slot, me, me_slot, tmvp, nullptr, md_out, works, results, 1);
}
extern "C" int svt_amd_md_encode_picture_inter16(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, const SvtAmdMdPicture *P, const SvtAmdMdInter *X, const SvtAmdMdLcu *lcus,
                                                 const uint16_t *src_y, uint32_t stride_y, const uint16_t *src_cb, const uint16_t *src_cr, uint32_t stride_c,
                                                 const SvtAmdOisLcuResult *ois, int ois_slot, const SvtAmdMeLcuResult *me, int me_slot, const SvtAmdTmvpLcu *tmvp,
                                                 SvtAmdMdLcuOut *md_out, SvtAmdLcuWork16 *works, SvtAmdLcuResult16 *results)
{
    if (!X)
        return SVT_AMD_ERR_BAD_PARAM;
    return md_encode_picture(ctx, pic, P, X, lcus, src_y, stride_y, src_cb, src_cr, stride_c, ois, ois_slot, me, me_slot, tmvp, nullptr, md_out, works, results, 2);
}

/* debug: stage clocks of the mode-decision kernel.  First call (out == NULL or not): switches the collection on for the picture object's later
 * calls; with out: 16 sums per LCU - [0] the LCU's surroundings into LDS, [1] lane 0: contexts + candidates, [2] the intra reference, [3] fast loop,
 * [4] lane 0: fast costs + candidate buffers, [5] full loop, [6] lane 0: costs + decision, [7] reconstruction + inter-depth decision, [8] neighbour
 * update + next unit, [9] state leaving LDS, [10] work record (+ merge / skip decisions), [11] encode pass, [12] waiting for neighbours, [15] calls */
extern "C" int svt_amd_debug_md_profile(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, unsigned long long *out)
{
    if (!ctx || !pic)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    SvtAmdMdState *m = nullptr;
    const int rc = md_state(pic, &m);
    if (rc)
        return rc;
    const size_t bytes = sizeof(unsigned long long) * 16 * (size_t)pic->nlcu;
    if (!m->d_prof) {
        HIP_TRY(hipMalloc((void **)&m->d_prof, 10 * bytes)); /* stage sums of every LCU, then sub-stage sums of every LCU, then both by unit depth (4 x 32 per LCU) */
        HIP_TRY(hipMemset(m->d_prof, 0, 10 * bytes));
    }
    if (out) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpy(out, m->d_prof, bytes, hipMemcpyDeviceToHost));
    }
    return SVT_AMD_OK;
}

/* measurement: duration (HIP events on the call's stream) and launch width of the picture object's last k_md_encode_picture launch */
extern "C" int svt_amd_debug_md_kernel_ms(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, float *ms, int *workgroups)
{
    if (!ctx || !pic || !pic->md || !ms)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(pic->md->ev_k1));
    HIP_TRY(hipEventElapsedTime(ms, pic->md->ev_k0, pic->md->ev_k1));
    if (workgroups)
        *workgroups = pic->md->grid;
    return SVT_AMD_OK;
}

/* measurement: duration of the encode-pass kernel that followed the picture object's last mode-decision launch (0 when the call asked for decisions only) */
extern "C" int svt_amd_debug_md_ep_ms(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, float *ms)
{
    if (!ctx || !pic || !pic->md || !ms)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(pic->md->ev_k2));
    HIP_TRY(hipEventElapsedTime(ms, pic->md->ev_k1, pic->md->ev_k2));
    return SVT_AMD_OK;
}

extern "C" int svt_amd_debug_md_flights(int *in_flight, int *workgroups_held, int *waiting)
{
    std::lock_guard<std::mutex> l(g_flight_mu);
    int fl = 0, wg = 0;
    for (const FlightDevice &fd : g_flight_dev)
        fl += fd.flights, wg += fd.wgs;
    if (in_flight)
        *in_flight = fl;
    if (workgroups_held)
        *workgroups_held = wg;
    if (waiting)
        *waiting = (int)g_flight_wait.size();
    return SVT_AMD_OK;
}
extern "C" int svt_amd_debug_md_kernel_lds_bytes(int inter, int bytes_per_sample)
{
    if (bytes_per_sample == 2)
        return (int)(inter ? sizeof(MdShared<true>) : sizeof(MdShared<false>)) + (int)sizeof(MdPictureDev);
    return (int)(inter ? sizeof(MdShared<true>) : sizeof(MdShared<false>)) + (int)sizeof(MdPictureDev);
}

/* debug: the finer marks of the mode-decision kernel (MD_SUB): 16 sums per LCU, collected together with svt_amd_debug_md_profile's (which switches the collection on) */
extern "C" int svt_amd_debug_md_profile_sub(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, unsigned long long *out)
{
    if (!ctx || !pic || !pic->md || !pic->md->d_prof || !out)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t bytes = sizeof(unsigned long long) * 16 * (size_t)pic->nlcu;
    HIP_TRY(hipMemcpy(out, pic->md->d_prof + 16 * (size_t)pic->nlcu, bytes, hipMemcpyDeviceToHost));
    return SVT_AMD_OK;
}

/* debug (-DMD_TRACE builds): lane-0 time stamps of the four waves along the unit chain of LCU `lcu`, units [unit, unit + 2) of its leaf list.  out == NULL: arms the picture
 * object's later calls; with out: [4 waves][1 + 128] words - the count, then (mark << 48 | shader clock) */
extern "C" int svt_amd_debug_md_trace(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, int lcu, int unit, unsigned long long *out)
{
#ifdef MD_TRACE
    if (!ctx || !pic)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    SvtAmdMdState *m = nullptr;
    const int rc = md_state(pic, &m);
    if (rc)
        return rc;
    const size_t bytes = sizeof(unsigned long long) * (4 * (1 + MD_TRACE_N) + 4);
    if (!m->d_trace)
        HIP_TRY(hipMalloc((void **)&m->d_trace, bytes));
    if (out) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpy(out, m->d_trace, bytes, hipMemcpyDeviceToHost));
    } else {
        HIP_TRY(hipMemset(m->d_trace, 0, bytes));
        m->trace_lcu = lcu, m->trace_unit = unit;
    }
    return SVT_AMD_OK;
#else
    (void)ctx, (void)pic, (void)lcu, (void)unit, (void)out;
    return SVT_AMD_ERR_BAD_PARAM;
#endif
}

/* debug: the picture object's later mode-decision calls run the 16x16 / 32x32 forward transforms of the full loops on the register butterflies (on != 0) instead of the
 * matrix cores - the path a unit outside the matrix form's wrap-free domain takes, which no 8-bit picture reaches by itself */
extern "C" int svt_amd_debug_md_force_butterflies(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, int on)
{
    if (!ctx || !pic)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    SvtAmdMdState *m = nullptr;
    const int rc = md_state(pic, &m);
    if (rc)
        return rc;
    m->force_butterflies = on != 0;
    return SVT_AMD_OK;
}

/* debug: the stage and sub-stage sums by the depth of the unit they were spent on: [LCU][depth 0..3][32] (slots 0..15 as svt_amd_debug_md_profile, 16..31 as _sub) */
extern "C" int svt_amd_debug_md_profile_depth(SvtAmdContext *ctx, SvtAmdEncDecPicture *pic, unsigned long long *out)
{
    if (!ctx || !pic || !pic->md || !pic->md->d_prof || !out)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out, pic->md->d_prof + 32 * (size_t)pic->nlcu, sizeof(unsigned long long) * 128 * (size_t)pic->nlcu, hipMemcpyDeviceToHost));
    return SVT_AMD_OK;
}

#ifdef EP_DEBUG_WINDOW_COUNTS
extern "C" __attribute__((visibility("default"))) int svt_amd_debug_window_counts(unsigned *out)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ep_dbg_counts), sizeof(unsigned) * 8));
    return SVT_AMD_OK;
}
#endif
