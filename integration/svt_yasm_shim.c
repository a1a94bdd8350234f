/*
 * integration/svt_yasm_shim.c - C stand-ins for the yasm routines of the reference, so that
 * /root/reference links on a machine without yasm / nasm (this image).  A maintainer with yasm
 * drops this file and assembles the reference's own .asm files instead.
 *
 * Compiled into integration/_build/ (the drop-in library) and, by oracle/Makefile, into
 * oracle/_ref/ (the reference build used as parity checker and CPU baseline).  It contains no
 * reference source; every symbol forwards to the reference's own C_DEFAULT peer.
 *
 * Symbols replaced (definitions in the reference):
 *   EbHevcLog2f_SSE2                    ASM_SSE2/EbPictureOperators_SSE2.asm:622 (bsr)
 *   PictureCopyKernel_SSE2              ASM_SSE2/EbPictureOperators_SSE2.asm
 *   ZeroOutCoeff{4x4_SSE,8x8,16x16,32x32}_SSE2   ASM_SSE2/EbPictureOperators_SSE2.asm
 *   GatherSaoStatisticsLcu16bit_SSE2    ASM_SSE2/EbGatherSaoStatistics16bit_SSE2.asm:200
 *   GatherSaoStatisticsLcu_OnlyEo_90_45_135_16bit_SSE2 (…asm:724)
 *   EbHevcRunEmms/SaveRegister/RestoreRegister   ASM_SSE2/x64RegisterUtil.asm:24-47
 * The two SAO forwards are "assumed equivalent" (SURVEY.md H5): unverifiable
 * without yasm.
 */
#include "EbDefinitions.h"
#include "EbPictureOperators_C.h"
#include "EbSampleAdaptiveOffset_C.h"

EB_U32 EbHevcLog2f_SSE2(EB_U32 x)
{
    /* bsr: index of the highest set bit; callers never pass 0 in practice,
       but bsr leaves the destination undefined there - return 0. */
    return x ? (EB_U32)(31 - __builtin_clz(x)) : 0;
}

void EbHevcRunEmms(void) {}
void EbHevcSaveRegister(void *p) { (void)p; }
void EbHevcRestoreRegister(void *p) { (void)p; }

void PictureCopyKernel_SSE2(EB_BYTE src, EB_U32 srcStride, EB_BYTE dst,
                            EB_U32 dstStride, EB_U32 areaWidth, EB_U32 areaHeight)
{
    PictureCopyKernel(src, srcStride, dst, dstStride, areaWidth, areaHeight, 1);
}

void PictureAverageKernel_SSE2(EB_BYTE src0, EB_U32 src0Stride, EB_BYTE src1,
                               EB_U32 src1Stride, EB_BYTE dst, EB_U32 dstStride,
                               EB_U32 areaWidth, EB_U32 areaHeight)
{
    PictureAverageKernel(src0, src0Stride, src1, src1Stride, dst, dstStride,
                         areaWidth, areaHeight);
}

#define ZERO_FWD(name)                                                        \
    void name(EB_S16 *coeffbuffer, EB_U32 coeffStride, EB_U32 coeffOriginIndex, \
              EB_U32 areaWidth, EB_U32 areaHeight)                            \
    {                                                                         \
        ZeroOutCoeffKernel(coeffbuffer, coeffStride, coeffOriginIndex,        \
                           areaWidth, areaHeight);                            \
    }
ZERO_FWD(ZeroOutCoeff4x4_SSE)
ZERO_FWD(ZeroOutCoeff8x8_SSE2)
ZERO_FWD(ZeroOutCoeff16x16_SSE2)
ZERO_FWD(ZeroOutCoeff32x32_SSE2)

EB_ERRORTYPE GatherSaoStatisticsLcu16bit_SSE2(
    EB_U16 *inputSamplePtr, EB_U32 inputStride, EB_U16 *reconSamplePtr,
    EB_U32 reconStride, EB_U32 lcuWidth, EB_U32 lcuHeight, EB_S32 *boDiff,
    EB_U16 *boCount, EB_S32 eoDiff[SAO_EO_TYPES][SAO_EO_CATEGORIES + 1],
    EB_U16 eoCount[SAO_EO_TYPES][SAO_EO_CATEGORIES + 1])
{
    return GatherSaoStatisticsLcu_62x62_16bit(inputSamplePtr, inputStride,
                                              reconSamplePtr, reconStride,
                                              lcuWidth, lcuHeight, boDiff,
                                              boCount, eoDiff, eoCount);
}

EB_ERRORTYPE GatherSaoStatisticsLcu_OnlyEo_90_45_135_16bit_SSE2(
    EB_U16 *inputSamplePtr, EB_U32 inputStride, EB_U16 *reconSamplePtr,
    EB_U32 reconStride, EB_U32 lcuWidth, EB_U32 lcuHeight,
    EB_S32 eoDiff[SAO_EO_TYPES][SAO_EO_CATEGORIES + 1],
    EB_U16 eoCount[SAO_EO_TYPES][SAO_EO_CATEGORIES + 1])
{
    return GatherSaoStatisticsLcu_62x62_OnlyEo_90_45_135_16bit(
        inputSamplePtr, inputStride, reconSamplePtr, reconStride, lcuWidth,
        lcuHeight, eoDiff, eoCount);
}
