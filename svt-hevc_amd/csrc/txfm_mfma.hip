/*
 * 32x32 forward / inverse DCT as INTEGER MFMA products (BASELINE.json north_star: "integer MFMA only for the 32x32 / 16x16 DCT
 * matrix products"): v_mfma_i32_32x32x32_i8 on gfx950.
 *
 * Replaces Transform32x32 / Transform32x32Estimate (C_DEFAULT/EbTransforms_C.c:1602, 1634: PartialButterfly32[Estimate] twice)
 * and InvTransform32x32 (:1910: PartialButterflyInverse32 twice).  The butterflies are exact integer evaluations of the matrix
 * products with the H.265 core matrix C (|c| <= 90: int8):
 *     forward   T1[i][j]  = (int16)((sum_k X[i][k] C[j][k] + 2^(s1-1)) >> s1)        out[l][j] = (int16)((sum_i C[l][i] T1[i][j] + 2^(s2-1)) >> s2)
 *     inverse   I1[r][n]  = clip16((sum_j Cf[j][r] C[j][n] + 64) >> 7)                res[r][n] = clip16((sum_j I1[j][r] C[j][n] + off) >> s2)
 * EXCEPT that the "Estimate" forward keeps its first two butterfly levels in 16 bits (EbTransforms_C.c:492-520): a sum of two
 * (level 1) or four (level 2) inputs that leaves [-32768, 32767] wraps there and not in a matrix product.  No wrap can happen while
 * every input of a pass is within +-8191 - which holds for every residual an 8-bit encode produces (|x| <= 255 gives |T1| <= 8160),
 * the only place the reference uses the Estimate form.  The kernel checks that bound per block and flags the blocks outside it;
 * the launcher re-runs exactly those through the VALU butterfly kernel (txfm_kernels.hip), so the entry point is bit-exact for
 * every int16 input.
 *
 * A 16-bit operand is fed to the int8 matrix cores as two byte planes:  x = 256 * hi + lo_u,  hi = x >> 8 (signed byte),
 * lo_u = x & 255 = (lo_u - 128) + 128 with (lo_u - 128) = lo_u ^ 0x80 a signed byte, so
 *     C . x = 256 (C . hi) + C . (lo_u ^ 0x80) + 128 * rowsum(C)         rowsum(C)[j] = 2048 for j == 0, else 0.
 * Two MFMAs per pass, four per block; the int32 accumulators hold the exact sums (|C . plane| <= 32 * 90 * 128).
 *
 * Register flow (one wave = one block at a time; D layout of the 32x32 MFMA: lane l, register v <-> row (v&3) + 8(v>>2) + 4(l>>5),
 * column l & 31): the pass-1 accumulator of lane (j, h) holds T1[row(v,h)][j], v = 0..15 - sixteen values along the index pass 2
 * contracts over - so it is byte-split in place and used directly as the next MFMA's operand; the constant operand is built with the
 * same slot order (the K index of an MFMA may be permuted freely as long as A and B agree).  No LDS, no cross-lane traffic.
 * The second product is computed transposed so that a lane ends up with four runs of four consecutive output samples (8-byte stores).
 */
#include "txfm_device.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define PERM(hi, lo, sel) __builtin_amdgcn_perm((uint32_t)(hi), (uint32_t)(lo), (uint32_t)(sel))
__device__ __forceinline__ int mfma_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

/* byte planes of four packed pairs: w[0..1] = four int16 -> one dword of high bytes, one of (low byte ^ 0x80) */
__device__ __forceinline__ void split_pairs(uint32_t w0, uint32_t w1, uint32_t &hi, uint32_t &lo)
{
    hi = PERM(w1, w0, 0x07050301u);
    lo = PERM(w1, w0, 0x06040200u) ^ 0x80808080u;
}
/* byte planes of sixteen int32 accumulator values already reduced to int16 range (low 16 bits significant) */
__device__ __forceinline__ void split_regs(const int (&t)[16], v4i &hi, v4i &lo)
{
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t p0 = PERM(t[4 * g + 1], t[4 * g + 0], 0x05040100u), p1 = PERM(t[4 * g + 3], t[4 * g + 2], 0x05040100u);
        uint32_t a, b;
        split_pairs(p0, p1, a, b);
        hi[g] = (int)a, lo[g] = (int)b;
    }
}
__device__ __forceinline__ int clip16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

/* constant operand: lane (m = l & 31, h = l >> 5) supplies, for slot e = 0..15, M[m][kidx(h, e)] (TRANSPOSED == false) or
 * M[kidx(h, e)][m] (true); kidx = 16 h + e (natural order: matches a lane's contiguous 16-sample load) or mfma_row(e, h) (the order
 * in which a lane holds the previous product's accumulator) */
template <bool TRANSPOSED, bool ACC_ORDER>
__device__ __forceinline__ v4i const_operand(int m, int h)
{
    v4i r;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int e = 4 * g + b, k = ACC_ORDER ? mfma_row(e, h) : 16 * h + e;
            const int8_t c = TRANSPOSED ? c_T32[k][m] : c_T32[m][k];
            w |= (uint32_t)(uint8_t)c << (8 * b);
        }
        r[g] = (int)w;
    }
    return r;
}

/* FORWARD.  kind 0: full precision; kind 1: Estimate (flags[b] = 1 when the block leaves the wrap-free domain). */
__global__ __launch_bounds__(256) void k_fwd32_mfma(const int16_t *__restrict__ src, int16_t *__restrict__ dst, uint32_t nblocks, int shift1,
                                                    int shift2, uint8_t *__restrict__ flags)
{
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    const v4i cB1 = const_operand<false, false>(m, h); /* pass 1, B: column j = m supplies C[j][16h + e]           */
    const v4i cB2 = const_operand<false, true>(m, h);  /* pass 2 (transposed product), B: column l = m supplies C[l][row(e, h)] */
    const int off1 = 1 << (shift1 - 1), off2 = 1 << (shift2 - 1);
    const int bias1 = m == 0 ? 128 * 2048 : 0; /* 128 * rowsum(C)[j]: pass-1 column j = m; pass-2 (transposed) column l = m */
    for (uint32_t b = wave; b < nblocks; b += nwaves) {
        /* A of pass 1: row i = m of the block, samples 16h .. 16h + 15 (32 contiguous bytes) */
        const uint4 *row = (const uint4 *)(src + (size_t)b * 1024 + m * 32 + 16 * h);
        const uint4 q0 = row[0], q1 = row[1];
        v4i ah, al;
        {
            uint32_t a, c;
            split_pairs(q0.x, q0.y, a, c), ah[0] = (int)a, al[0] = (int)c;
            split_pairs(q0.z, q0.w, a, c), ah[1] = (int)a, al[1] = (int)c;
            split_pairs(q1.x, q1.y, a, c), ah[2] = (int)a, al[2] = (int)c;
            split_pairs(q1.z, q1.w, a, c), ah[3] = (int)a, al[3] = (int)c;
        }
        int viol = 0;
        if (flags) { /* Estimate: |x| <= 8191 <=> the high byte is within [-32, 31] ... checked exactly on the 16-bit values */
            const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int lo16 = (int16_t)(w[k] & 0xffffu), hi16 = (int)w[k] >> 16;
                viol |= (lo16 > 8191) | (lo16 < -8191) | (hi16 > 8191) | (hi16 < -8191);
            }
        }
        v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const v16i dh = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah, cB1, z, 0, 0, 0);
        const v16i dl = __builtin_amdgcn_mfma_i32_32x32x32_i8(al, cB1, z, 0, 0, 0);
        /* T1[row(v,h)][j = m] */
        int t1[16];
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const int s = (dh[v] << 8) + dl[v] + bias1;
            t1[v] = (int)(int16_t)((s + off1) >> shift1);
            if (flags)
                viol |= (t1[v] > 8191) | (t1[v] < -8191);
        }
        /* pass 2, transposed: out^T[j][l] = sum_i T1[i][j] C[l][i]:  A = T1 (row j = m, slots = this lane's registers), B = C */
        v4i th, tl;
        split_regs(t1, th, tl);
        const v16i eh = __builtin_amdgcn_mfma_i32_32x32x32_i8(th, cB2, z, 0, 0, 0);
        const v16i el = __builtin_amdgcn_mfma_i32_32x32x32_i8(tl, cB2, z, 0, 0, 0);
        /* lane (l = m, h) holds out[l][j = row(v, h)]: four runs of four consecutive samples */
        int16_t *orow = dst + (size_t)b * 1024 + m * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            int o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int s = (eh[4 * g + k] << 8) + el[4 * g + k] + bias1;
                o[k] = (int)(int16_t)((s + off2) >> shift2);
            }
            uint2 st;
            st.x = PERM(o[1], o[0], 0x05040100u), st.y = PERM(o[3], o[2], 0x05040100u);
            *(uint2 *)(orow + 8 * g) = st;
        }
        if (flags) {
            const unsigned long long any = __ballot(viol != 0);
            if (lane == 0)
                flags[b] = any != 0;
        }
    }
}

/* INVERSE (InvTransform32x32): stage 1 contracts over the ROW index of the coefficient block, so lane (r = m, h) gathers column r. */
__global__ __launch_bounds__(256) void k_inv32_mfma(const int16_t *__restrict__ src, int16_t *__restrict__ dst, uint32_t nblocks, int shift1,
                                                    int shift2)
{
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    const v4i cB1 = const_operand<true, false>(m, h); /* stage 1, B: column n = m supplies C[16h + e][n]           */
    const v4i cA2 = const_operand<true, true>(m, h);  /* stage 2 (transposed), A: row n2 = m supplies C[row(e,h)][n2] */
    const int off1 = 1 << (shift1 - 1), off2 = 1 << (shift2 - 1);
    for (uint32_t b = wave; b < nblocks; b += nwaves) {
        /* A of stage 1: I1[r][n] = sum_j Cf[j][r] C[j][n]: row r = m supplies Cf[16h + e][r] (a column of the block) */
        const int16_t *col = src + (size_t)b * 1024 + (16 * h) * 32 + m;
        int cf[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
            cf[e] = col[e * 32];
        v4i ah, al;
        split_regs(cf, ah, al);
        v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const v16i dh = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah, cB1, z, 0, 0, 0);
        const v16i dl = __builtin_amdgcn_mfma_i32_32x32x32_i8(al, cB1, z, 0, 0, 0);
        /* lane (n = m, h) holds I1[r = row(v,h)][n]; bias: 128 * sum_j C[j][n] over the slots = 128 * column sum of C */
        int colsum = 0;
#pragma unroll
        for (int j = 0; j < 32; j++)
            colsum += c_T32[j][m];
        int i1[16];
#pragma unroll
        for (int v = 0; v < 16; v++)
            i1[v] = clip16(((dh[v] << 8) + dl[v] + 128 * colsum + off1) >> shift1);
        /* stage 2: res[r2][n2] = sum_j I1[j][r2] C[j][n2]; transposed product D[n2][r2]: A = C^T rows n2 (constant), B = I1 (column r2 =
         * this lane's m, slots = its registers: I1[row(e,h)][m]) */
        v4i th, tl;
        split_regs(i1, th, tl);
        const v16i eh = __builtin_amdgcn_mfma_i32_32x32x32_i8(cA2, th, z, 0, 0, 0);
        const v16i el = __builtin_amdgcn_mfma_i32_32x32x32_i8(cA2, tl, z, 0, 0, 0);
        /* lane (r2 = m, h) holds res[r2][n2 = row(v,h)]; bias depends on the ROW n2 of D: 128 * sum_j C[j][n2] */
        int16_t *orow = dst + (size_t)b * 1024 + m * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            int o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int n2 = 8 * g + 4 * h + k;
                int cs = 0;
#pragma unroll
                for (int j = 0; j < 32; j++)
                    cs += c_T32[j][n2];
                o[k] = clip16(((eh[4 * g + k] << 8) + el[4 * g + k] + 128 * cs + off2) >> shift2);
            }
            uint2 st;
            st.x = PERM(o[1], o[0], 0x05040100u), st.y = PERM(o[3], o[2], 0x05040100u);
            *(uint2 *)(orow + 8 * g) = st;
        }
    }
}

/* VALU butterfly kernel of txfm_kernels.hip over the flagged blocks only */
int svt_amd_launch_fwd_transform_flagged(hipStream_t st, int kind, int size, uint32_t inc, const int16_t *d_res, int16_t *d_coeff, uint32_t n,
                                         const uint8_t *d_only);

extern "C" int svt_amd_fwd_transform_mfma_batch(SvtAmdContext *ctx, int kind, int size, uint32_t bitIncrement, const int16_t *d_residual,
                                                int16_t *d_coeff, uint32_t nblocks)
{
    if (!ctx || !d_residual || !d_coeff || !nblocks || size != 32 || (kind != 0 && kind != 1) || bitIncrement > 4)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    /* Transform32x32: 4 + inc / 11; Transform32x32Estimate: 6 + inc / 9 (C_DEFAULT/EbTransforms_C.c:1611, 1644) */
    const int s1 = (kind ? 6 : 4) + (int)bitIncrement, s2 = kind ? 9 : 11;
    uint8_t *flags = nullptr;
    if (kind == 1) {
        int rc = svt_amd_ctx_scratch(ctx, nblocks, &flags);
        if (rc)
            return rc;
    }
    const unsigned grid = (nblocks + 3) / 4 < 2048 ? (nblocks + 3) / 4 : 2048;
    hipLaunchKernelGGL(k_fwd32_mfma, dim3(grid), dim3(256), 0, ctx->stream, d_residual, d_coeff, nblocks, s1, s2, flags);
    HIP_TRY(hipGetLastError());
    if (kind == 1)
        return svt_amd_launch_fwd_transform_flagged(ctx->stream, kind, size, bitIncrement, d_residual, d_coeff, nblocks, flags);
    return SVT_AMD_OK;
}

extern "C" int svt_amd_inv_transform_mfma_batch(SvtAmdContext *ctx, int size, uint32_t bitIncrement, const int16_t *d_coeff, int16_t *d_residual,
                                                uint32_t nblocks)
{
    if (!ctx || !d_residual || !d_coeff || !nblocks || size != 32 || bitIncrement > 4)
        return SVT_AMD_ERR_BAD_PARAM;
    HIP_TRY(hipSetDevice(ctx->device));
    const unsigned grid = (nblocks + 3) / 4 < 2048 ? (nblocks + 3) / 4 : 2048;
    hipLaunchKernelGGL(k_inv32_mfma, dim3(grid), dim3(256), 0, ctx->stream, d_coeff, d_residual, nblocks, 7, 12 - (int)bitIncrement);
    HIP_TRY(hipGetLastError());
    return SVT_AMD_OK;
}
