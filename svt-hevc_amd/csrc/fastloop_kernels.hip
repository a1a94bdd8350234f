/*
 * Distortion stage of the mode decision's fast loop (EncDec row "ProductPerformFastLoop + NxMSadKernel").
 *
 * ProductPerformFastLoop (Codec/EbProductCodingLoop.c:1911-2190) predicts every candidate of the second fast-cost search into
 * its buffer and measures it against the source block: luma SAD (NxMSadKernel_funcPtrArray[..][cuSize >> 3], :2044-2051) and,
 * with useChromaInformationInFastLoop, Cb SAD + Cr SAD (:2054-2078); candidates flagged as most-probable-mode skip the
 * measurement (:2040).  Here the predictions are already in HBM (svt_amd_intra_pu_batch / svt_amd_inter_pu_batch wrote them),
 * so the stage is one launch over the candidate list: one wave per candidate, v_sad_u8 on packed words, shuffle reduction.
 * HBM bytes per candidate: 2 * size^2 (+ size^2 with chroma) in, 8 out.
 */
#include "leaf_util.h"

struct FastLoopCand { int32_t src_off_y, src_off_c, pred_off_y, pred_off_c; uint8_t size, flags, pad[2]; }; /* = SvtAmdFastLoopCand */
struct FastLoopDist { uint32_t luma, chroma; };                                                            /* = SvtAmdFastLoopDist */

__device__ __forceinline__ uint32_t block_sad(const uint8_t *__restrict__ a, int sa, const uint8_t *__restrict__ b, int sb, int n, int lane)
{
    uint32_t s = 0;
    if ((((uintptr_t)a | (uintptr_t)b | (uint32_t)sa | (uint32_t)sb) & 3) == 0) {
        const int wpr = n >> 2; /* words per row */
        for (int i = lane; i < wpr * n; i += 64) {
            const int y = i / wpr, x = i - y * wpr;
            s = __builtin_amdgcn_sad_u8(*(const uint32_t *)(a + (size_t)y * sa + 4 * x), *(const uint32_t *)(b + (size_t)y * sb + 4 * x), s);
        }
    } else {
        for (int i = lane; i < n * n; i += 64) {
            const int y = i / n, x = i - y * n;
            s += (uint32_t)abs((int)a[(size_t)y * sa + x] - (int)b[(size_t)y * sb + x]);
        }
    }
    for (int o = 32; o > 0; o >>= 1)
        s += __shfl_xor(s, o);
    return s;
}

__global__ void __launch_bounds__(256) k_fast_loop_distortion(const uint8_t *__restrict__ srcY, int srcStrideY, const uint8_t *__restrict__ srcCb,
                                                              const uint8_t *__restrict__ srcCr, int srcStrideC,
                                                              const uint8_t *__restrict__ predY, int predStrideY,
                                                              const uint8_t *__restrict__ predCb, const uint8_t *__restrict__ predCr,
                                                              int predStrideC, const FastLoopCand *__restrict__ cands, uint32_t ncand,
                                                              FastLoopDist *__restrict__ out)
{
    const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= ncand)
        return;
    const FastLoopCand K = cands[c];
    uint32_t luma = 0, chroma = 0;
    if (!(K.flags & 2)) { /* not a most-probable-mode candidate */
        luma = block_sad(srcY + K.src_off_y, srcStrideY, predY + K.pred_off_y, predStrideY, K.size, lane);
        if (K.flags & 1) {
            chroma = block_sad(srcCb + K.src_off_c, srcStrideC, predCb + K.pred_off_c, predStrideC, K.size >> 1, lane);
            chroma += block_sad(srcCr + K.src_off_c, srcStrideC, predCr + K.pred_off_c, predStrideC, K.size >> 1, lane);
        }
    }
    if (lane == 0)
        out[c] = {luma, chroma};
}

extern "C" int svt_amd_fast_loop_distortion_batch(SvtAmdContext *ctx, const uint8_t *d_src_y, uint32_t srcStrideY, const uint8_t *d_src_cb,
                                                  const uint8_t *d_src_cr, uint32_t srcStrideC, const uint8_t *d_pred_y,
                                                  uint32_t predStrideY, const uint8_t *d_pred_cb, const uint8_t *d_pred_cr,
                                                  uint32_t predStrideC, const SvtAmdFastLoopCand *d_cands, uint32_t ncand,
                                                  SvtAmdFastLoopDist *d_out)
{
    static_assert(sizeof(FastLoopCand) == sizeof(SvtAmdFastLoopCand) && sizeof(FastLoopDist) == sizeof(SvtAmdFastLoopDist), "layout");
    if (!ctx || !d_src_y || !d_pred_y || !d_cands || !d_out || !ncand || !srcStrideY || !predStrideY ||
        ((d_src_cb || d_src_cr || d_pred_cb || d_pred_cr) && !(d_src_cb && d_src_cr && d_pred_cb && d_pred_cr && srcStrideC && predStrideC)))
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_fast_loop_distortion, dim3((ncand + 3) / 4), dim3(256), 0, ctx->stream, d_src_y, (int)srcStrideY, d_src_cb, d_src_cr,
                       (int)srcStrideC, d_pred_y, (int)predStrideY, d_pred_cb, d_pred_cr, (int)predStrideC, (const FastLoopCand *)d_cands, ncand,
                       (FastLoopDist *)d_out);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
