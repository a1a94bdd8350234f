/*
 * oracle/ref_harness_coeffscan.c - TEST INFRASTRUCTURE ONLY; compiled INTO oracle/_ref/libsvtref.so next to the reference's own objects.
 * One arithmetic-coder state of the reference (CabacEncodeContext_t + OutputBitstreamUnit_t, Codec/EbEntropyCodingUtil.h:176-190) that tests feed
 * transform block by transform block, either through the reference's own EncodeQuantizedCoefficients_generic / _SSE2 (Codec/EbEntropyCoding.c:1172 /
 * :1716) on the s16 coefficients, or through the pre-scan consumer (integration/svt_coeff_scan_consumer.h) on the pre-scan records; svt_ref_cabac_state
 * serialises everything the coder has produced and holds (bytes written, byte holder, interval, context models) so the two runs can be compared.
 */
#include <stdlib.h>
#include <string.h>
#include "EbEntropyCoding.h"
#include "EbEntropyCodingUtil.h"
#include "EbBitstreamUnit.h"
#include "EbTransformUnit.h"
#include "../integration/svt_coeff_scan_consumer.h"

typedef struct {
    CabacEncodeContext_t cabac;
    OutputBitstreamUnit_t bs;
} RefCabac;

void *svt_ref_cabac_new(uint32_t seed)
{
    RefCabac *h = (RefCabac *)calloc(1, sizeof(*h));
    if (!h || OutputBitstreamUnitCtor(&h->bs, 1u << 22) != EB_ErrorNone)
        return NULL;
    h->cabac.bacEncContext.m_pcTComBitIf = &h->bs;
    h->cabac.bacEncContext.intervalLowValue = 0, h->cabac.bacEncContext.intervalRangeValue = 510; /* ResetBacEnc, EbEntropyCoding.c:100 */
    h->cabac.bacEncContext.bitsRemainingNum = 23, h->cabac.bacEncContext.tempBufferedBytesNum = 0, h->cabac.bacEncContext.tempBufferedByte = 0xff;
    h->cabac.colorFormat = EB_YUV420;
    EB_ContextModel *m = (EB_ContextModel *)&h->cabac.contextModelEncContext;
    for (size_t i = 0; i < sizeof(h->cabac.contextModelEncContext) / sizeof(EB_ContextModel); i++) {
        seed = seed * 1664525u + 1013904223u;
        m[i] = (seed >> 16) % 126; /* (state << 1) | mps, states 0..62 */
    }
    return h;
}

void svt_ref_cabac_free(void *hv)
{
    RefCabac *h = (RefCabac *)hv;
    if (h) {
        free(h->bs.bufferBegin);
        free(h);
    }
}

/* the reference's coder on the coefficients of one transform block; asm_form 0: _generic, 1: the table's SIMD form */
void svt_ref_cabac_code_raw(void *hv, uint32_t size, uint32_t type, uint32_t intra_luma_mode, int16_t *coeff, uint32_t stride, uint32_t component,
                            uint32_t nz, int asm_form)
{
    RefCabac *h = (RefCabac *)hv;
    TransformUnit_t tu;
    memset(&tu, 0, sizeof(tu));
    tu.nzCoefCount[component] = (EB_U16)nz; /* component: plane 0 Y / 1 Cb / 2 Cr */
    (asm_form ? EncodeQuantizedCoefficients_SSE2 : EncodeQuantizedCoefficients_generic)(&h->cabac, size, (EB_MODETYPE)type, intra_luma_mode, EB_INTRA_CHROMA_DM,
                                                                                       coeff, stride, component == 0 ? COMPONENT_LUMA : component == 1 ? COMPONENT_CHROMA_CB : COMPONENT_CHROMA_CR, &tu);
}

void svt_ref_cabac_code_scan(void *hv, uint32_t size, uint32_t component, const SvtAmdCoeffScanTu *tu, const SvtAmdCoeffScanGroup *groups,
                             const uint16_t *levels)
{
    RefCabac *h = (RefCabac *)hv;
    svt_coeff_scan_encode(&h->cabac, size, component != 0, tu, groups, levels);
}

uint32_t svt_ref_cabac_state(void *hv, uint8_t *out, uint32_t cap)
{
    RefCabac *h = (RefCabac *)hv;
    const uint32_t words = (uint32_t)(h->bs.buffer - h->bs.bufferBegin);
    const uint32_t need = 4 * words + 8 + 20 + (uint32_t)sizeof(h->cabac.contextModelEncContext);
    if (need > cap)
        return 0;
    uint8_t *p = out;
    memcpy(p, h->bs.bufferBegin, 4 * words), p += 4 * words;
    memcpy(p, &h->bs.byteHolder, 4), p += 4;
    memcpy(p, &h->bs.validBitsCount, 4), p += 4;
    memcpy(p, &h->cabac.bacEncContext.intervalLowValue, 20), p += 20; /* low, range, buffered byte, buffered bytes, bits remaining */
    memcpy(p, &h->cabac.contextModelEncContext, sizeof(h->cabac.contextModelEncContext));
    return need;
}
