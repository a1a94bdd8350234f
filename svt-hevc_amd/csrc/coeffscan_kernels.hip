/*
 * coeffscan_kernels.hip - entropy hand-off pre-scan (include/svt_hevc_amd.h "Entropy hand-off pre-scan").
 *
 * What EncodeQuantizedCoefficients (Codec/EbEntropyCoding.c:1172) does with a transform block before its first bin - mode-dependent scan
 * :1347-1372, sub-block / scan re-ordering with one significance map per 4x4 sub-block :1378-1430, last significant position :1432-1461, the
 * DC-only fast track :1308 - and the sign / greater-1 words its level loop derives from the same values (:1532-1600), for EVERY transform block of
 * a picture in one launch, from the encode pass's records where they lie in HBM.  The host's CABAC loop (integration/svt_coeff_scan_consumer.h) then
 * reads 8 bytes per block, 8 per coded sub-block and 2 per non-zero coefficient instead of the s16 planes.
 *
 * k_coeff_scan: one 256-thread workgroup per LCU.  A THREAD owns a 4x4 sub-block (an LCU has at most 384 of them over its three planes: two
 * passes): four 8-byte loads, the 16 values stay in registers; significance map, count, sign and greater-1 words are bit operations on them.
 * What needs the whole block (last sub-block, where its groups and levels start) is three small exclusive scans over the LCU's <= 192 blocks
 * in LDS.  Groups and levels go to per-LCU pools; k_coeff_scan_bases (one workgroup) turns the per-LCU counts into picture offsets and
 * k_coeff_scan_compact packs the pools in LCU order, so the host copies exactly what was produced.
 * Bound: HBM - the s16 planes are read once (3 * W * H bytes at 4:2:0), the output is a few percent of that.
 */
#include <stddef.h>
#include "rate_device.h"
#include "svt_amd_internal.h"

#define CS_MAX_TUS (3 * SVT_AMD_LCU_MAX_CUS)
#define CS_MAX_SUBS 384
#define CS_MAX_LEVELS 6144

struct CsTu {
    uint16_t coeff_off;   /* first coefficient in its plane (LCU-local) */
    uint8_t plane, lg, scan, valid;
    uint16_t nz;          /* the unit's recorded count (nzCoefCount) */
    uint16_t sub_base;    /* first sub-block in the LCU's sub-block index space */
    int16_t last;
    uint16_t first_group, level_base;
};

struct CsShared {
    CsTu tu[CS_MAX_TUS];
    uint16_t scan_in[256];                  /* scratch of the block-wide scans */
    uint16_t wave_tot[4];
    uint8_t owner[CS_MAX_SUBS];             /* block of a sub-block index */
    uint16_t sig[CS_MAX_SUBS];
    uint8_t cnt[CS_MAX_SUBS];
    uint16_t lvl_off[CS_MAX_SUBS];          /* first level of the sub-block in the LCU's list */
    int16_t grp[CS_MAX_SUBS];               /* its group in the LCU's list, -1 = beyond the last sub-block */
    uint32_t total_subs, total_groups, total_levels;
};

/* exclusive scan of one value per thread over the 256 threads; returns this thread's prefix, *total = sum */
__device__ __forceinline__ uint32_t cs_scan256(CsShared &S, uint32_t v, uint32_t *total)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= d)
            x += y;
    }
    if (lane == 63)
        S.wave_tot[wave] = (uint16_t)x;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (w < wave)
            base += S.wave_tot[w];
        sum += S.wave_tot[w];
    }
    __syncthreads();
    *total = sum;
    return base + x - v;
}

/* the 16 values of a sub-block in forward scan order k = 0..15: which raster position (y * 4 + x) scan position k reads.  Diagonal: up-right
 * diagonals (H.265 6.5.3); horizontal: raster; vertical: column by column - the reference's scans4 tables with its x / y swap of SCAN_HOR2 folded
 * in (:1398-1410). */
__device__ __forceinline__ int cs_raster_of(int scan, int k)
{
    constexpr uint8_t diag[16] = {0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15};
    return scan == 0 ? diag[k] : scan == 1 ? k : ((k & 3) << 2) | (k >> 2);
}

struct CsSub {
    uint32_t sig, cnt, sign, gt1;
    uint16_t level[16]; /* coding order; [0, cnt) valid */
};

template <int SCAN>
__device__ __forceinline__ void cs_digest(const int16_t (&v)[16], CsSub &o)
{
    o.sig = 0, o.cnt = 0, o.sign = 0, o.gt1 = 0;
#pragma unroll
    for (int k = 0; k < 16; k++)
        o.sig |= (uint32_t)(v[cs_raster_of(SCAN, k)] != 0) << k;
    /* coded coefficients: the set positions from the top down */
#pragma unroll
    for (int i = 0; i < 16; i++)
        o.level[i] = 0;
#pragma unroll
    for (int k = 15; k >= 0; k--) {
        const int c = v[cs_raster_of(SCAN, k)];
        if (c != 0) {
            const uint32_t a = (uint32_t)(c < 0 ? -c : c);
            o.sign = o.sign * 2 + (c < 0);
            o.gt1 |= (uint32_t)(a > 1) << o.cnt;
            /* level[cnt] = a without a dynamically indexed register array: a select per slot */
#pragma unroll
            for (int i = 0; i < 16; i++)
                o.level[i] = (uint32_t)i == o.cnt ? (uint16_t)a : o.level[i];
            o.cnt++;
        }
    }
}

__global__ __launch_bounds__(256) void k_coeff_scan(const uint8_t *__restrict__ works, size_t work_stride, const uint8_t *__restrict__ results,
                                                    size_t result_stride, SvtAmdCoeffScanLcu *__restrict__ lcus,
                                                    SvtAmdCoeffScanGroup *__restrict__ group_pool, uint16_t *__restrict__ level_pool)
{
    __shared__ CsShared S;
    const int t = threadIdx.x, lcu = blockIdx.x;
    const SvtAmdLcuWork *W = (const SvtAmdLcuWork *)(works + (size_t)lcu * work_stride);
    const SvtAmdLcuResult *R = (const SvtAmdLcuResult *)(results + (size_t)lcu * result_stride);
    /* ---- 1. the LCU's transform blocks: block i = 3 * slot + plane ---- */
    uint32_t nsub = 0;
    if (t < CS_MAX_TUS) {
        const int c = t / 3, p = t - 3 * c;
        const int ncu = W->num_cus;
        const bool big = ncu == 1 && W->cu[0].size == 64;
        CsTu T;
        T.coeff_off = 0, T.plane = (uint8_t)p, T.lg = 2, T.scan = 0, T.valid = 0, T.nz = 0, T.sub_base = 0, T.last = -1, T.first_group = 0, T.level_base = 0;
        const bool exists = big ? (c >= 1 && c <= 4) : c < ncu;
        if (exists && R->cu[c].cbf[p]) {
            const SvtAmdLcuCu &u = W->cu[big ? 0 : c];
            const uint32_t size = big ? 32u : u.size, x = big ? 32u * ((c - 1) & 1) : u.x, y = big ? 32u * ((c - 1) >> 1) : u.y;
            const uint32_t ts = p ? (size == 8 ? 4u : size >> 1) : size;
            T.lg = (uint8_t)(31 - __clz((int)ts));
            T.coeff_off = (uint16_t)(p ? 32 * (y >> 1) + (x >> 1) : 64 * y + x);
            T.valid = 1, T.nz = R->cu[c].nz[p];
            if (u.pred_mode == 2 && T.lg <= 3 - (p != 0)) { /* :1347: mode-dependent scan of small intra blocks (chroma follows the luma mode: DM) */
                const int m = u.intra_luma_mode;
                int d = 8 - ((m - 2) & 15);
                d = d < 0 ? -d : d;
                if (d <= 4)
                    T.scan = (m & 16) ? 1 : 2;
            }
            nsub = 1u << (2 * (T.lg - 2));
        }
        S.tu[t] = T;
    }
    uint32_t total_subs;
    const uint32_t sub_base = cs_scan256(S, nsub, &total_subs);
    if (t < CS_MAX_TUS && nsub) {
        S.tu[t].sub_base = (uint16_t)sub_base;
        for (uint32_t j = 0; j < nsub; j++)
            S.owner[sub_base + j] = (uint8_t)t;
    }
    __syncthreads();
    /* ---- 2. a thread per sub-block: values into registers, maps / counts into LDS ---- */
    CsSub sub[2];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t q = (uint32_t)t + 256u * pass;
        sub[pass].sig = 0, sub[pass].cnt = 0, sub[pass].sign = 0, sub[pass].gt1 = 0;
        if (q < total_subs) {
            const CsTu T = S.tu[S.owner[q]];
            const uint32_t s = q - T.sub_base;
            uint32_t gy = c_rt.sb[T.lg - 2][s] >> 4, gx = c_rt.sb[T.lg - 2][s] & 15;
            if (T.scan == 1) { const uint32_t tmp = gx; gx = gy, gy = tmp; } /* sub-block scan mirrored for the horizontal scan (:1388) */
            const uint32_t stride = T.plane ? 32 : 64;
            const int16_t *plane = T.plane == 0 ? R->coeff_y : T.plane == 1 ? R->coeff_cb : R->coeff_cr;
            const int16_t *p0 = plane + T.coeff_off + 4 * gy * stride + 4 * gx;
            int16_t v[16];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint2 w = *(const uint2 *)(p0 + r * stride); /* 4 s16, 8-byte aligned: x and the plane offsets are multiples of 4 */
                v[4 * r + 0] = (int16_t)(w.x & 0xFFFF), v[4 * r + 1] = (int16_t)(w.x >> 16);
                v[4 * r + 2] = (int16_t)(w.y & 0xFFFF), v[4 * r + 3] = (int16_t)(w.y >> 16);
            }
            if (T.scan == 0)
                cs_digest<0>(v, sub[pass]);
            else if (T.scan == 1)
                cs_digest<1>(v, sub[pass]);
            else
                cs_digest<2>(v, sub[pass]);
            S.sig[q] = (uint16_t)sub[pass].sig, S.cnt[q] = (uint8_t)sub[pass].cnt;
        }
    }
    __syncthreads();
    /* ---- 3. per block: last sub-block, number of groups and levels ---- */
    uint32_t ngroups = 0, nlevels = 0;
    if (t < CS_MAX_TUS && S.tu[t].valid) {
        const CsTu T = S.tu[t];
        const uint32_t n = 1u << (2 * (T.lg - 2));
        int last = -1;
        for (uint32_t s = 0; s < n; s++)
            if (S.sig[T.sub_base + s])
                last = (int)s, nlevels += S.cnt[T.sub_base + s];
        S.tu[t].last = (int16_t)last;
        ngroups = (uint32_t)(last + 1);
    }
    uint32_t total_groups, total_levels;
    const uint32_t first_group = cs_scan256(S, ngroups, &total_groups);
    const uint32_t level_base = cs_scan256(S, nlevels, &total_levels);
    if (t < CS_MAX_TUS) {
        const CsTu T = S.tu[t];
        SvtAmdCoeffScanTu o;
        o.scan_index = 0, o.last_scan_set = -1, o.pos_last = 0, o.last_x = 0, o.last_y = 0, o.dc_only = 0, o.first_group = 0;
        if (T.valid) {
            const uint32_t n = 1u << (2 * (T.lg - 2));
            const int last = T.last;
            o.first_group = (uint16_t)first_group, o.last_scan_set = (int8_t)last;
            o.dc_only = T.nz == 1 && (S.sig[T.sub_base] & 1); /* :1308 */
            o.scan_index = o.dc_only ? 0 : T.scan;
            uint32_t lv = level_base;
            for (int s = (int)n - 1; s >= 0; s--) { /* coding order: from the last sub-block down */
                S.grp[T.sub_base + s] = (int16_t)(s <= last ? (int)first_group + (last - s) : -1);
                S.lvl_off[T.sub_base + s] = (uint16_t)lv;
                if (s <= last)
                    lv += S.cnt[T.sub_base + s];
            }
            if (last >= 0 && !o.dc_only) {
                const uint32_t pos_last = 31u - (uint32_t)__clz((int)S.sig[T.sub_base + last]); /* :1444 */
                uint32_t ly = 4u * (c_rt.sb[T.lg - 2][last] >> 4), lx = 4u * (c_rt.sb[T.lg - 2][last] & 15);
                const uint32_t pl = T.scan ? c_rt.col4[pos_last] : c_rt.diag4[pos_last];
                ly += pl >> 2, lx += pl & 3;
                if (T.scan) { const uint32_t tmp = lx; lx = ly, ly = tmp; }
                o.pos_last = (uint8_t)pos_last, o.last_x = (uint8_t)lx, o.last_y = (uint8_t)ly;
            }
        }
        lcus[lcu].tu[T.plane][t / 3] = o;
    }
    if (t == 0) {
        lcus[lcu].group_base = 0, lcus[lcu].level_base = 0;
        lcus[lcu].groups = (uint16_t)total_groups, lcus[lcu].levels = (uint16_t)total_levels;
        lcus[lcu].pad[0] = lcus[lcu].pad[1] = lcus[lcu].pad[2] = lcus[lcu].pad[3] = 0;
    }
    __syncthreads();
    /* ---- 4. the sub-block threads write their group and levels into the LCU's pools ---- */
    SvtAmdCoeffScanGroup *gp = group_pool + (size_t)lcu * CS_MAX_SUBS;
    uint16_t *lp = level_pool + (size_t)lcu * CS_MAX_LEVELS;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t q = (uint32_t)t + 256u * pass;
        if (q < total_subs && S.grp[q] >= 0) {
            SvtAmdCoeffScanGroup g;
            g.sigmap = (uint16_t)sub[pass].sig, g.sign = (uint16_t)sub[pass].sign, g.gt1 = (uint16_t)sub[pass].gt1, g.first_level = S.lvl_off[q];
            gp[S.grp[q]] = g;
#pragma unroll
            for (int i = 0; i < 16; i++)
                if ((uint32_t)i < sub[pass].cnt)
                    lp[S.lvl_off[q] + i] = sub[pass].level[i];
        }
    }
}

/* picture offsets of the LCUs' lists: one workgroup, LCUs in chunks of 1024 */
__global__ __launch_bounds__(1024) void k_coeff_scan_bases(SvtAmdCoeffScanLcu *lcus, int n, uint32_t *totals)
{
    __shared__ uint32_t wt[2][16];
    __shared__ uint32_t carry[2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t < 2)
        carry[t] = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + t;
        uint32_t v[2] = {i < n ? lcus[i].groups : 0u, i < n ? lcus[i].levels : 0u}, x[2] = {v[0], v[1]};
#pragma unroll
        for (int k = 0; k < 2; k++) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x[k], d);
                if (lane >= d)
                    x[k] += y;
            }
            if (lane == 63)
                wt[k][wave] = x[k];
        }
        __syncthreads();
        uint32_t pre[2] = {carry[0], carry[1]}, sum[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 2; k++)
            for (int w = 0; w < 16; w++) {
                if (w < wave)
                    pre[k] += wt[k][w];
                sum[k] += wt[k][w];
            }
        if (i < n)
            lcus[i].group_base = pre[0] + x[0] - v[0], lcus[i].level_base = pre[1] + x[1] - v[1];
        __syncthreads();
        if (t < 2)
            carry[t] += sum[t];
        __syncthreads();
    }
    if (t < 2)
        totals[t] = carry[t];
}

__global__ __launch_bounds__(256) void k_coeff_scan_compact(const SvtAmdCoeffScanLcu *__restrict__ lcus, const SvtAmdCoeffScanGroup *__restrict__ group_pool,
                                                            const uint16_t *__restrict__ level_pool, SvtAmdCoeffScanGroup *__restrict__ groups,
                                                            uint32_t group_capacity, uint16_t *__restrict__ levels, uint32_t level_capacity)
{
    const int lcu = blockIdx.x;
    const uint32_t gb = lcus[lcu].group_base, lb = lcus[lcu].level_base, ng = lcus[lcu].groups, nl = lcus[lcu].levels;
    if (gb + ng <= group_capacity)
        for (uint32_t i = threadIdx.x; i < ng; i += 256)
            groups[gb + i] = group_pool[(size_t)lcu * CS_MAX_SUBS + i];
    if (lb + nl <= level_capacity)
        for (uint32_t i = threadIdx.x; i < nl; i += 256)
            levels[lb + i] = level_pool[(size_t)lcu * CS_MAX_LEVELS + i];
}

extern "C" int svt_amd_coeff_scan_picture(SvtAmdContext *ctx, const void *works, size_t work_stride, const void *results, size_t result_stride,
                                          int device_arrays, int n_lcus, SvtAmdCoeffScanLcu *lcus, SvtAmdCoeffScanGroup *groups, uint32_t group_capacity,
                                          uint16_t *levels, uint32_t level_capacity, uint32_t totals[2])
{
    const size_t work_head = offsetof(SvtAmdLcuWork, src_y), result_head = offsetof(SvtAmdLcuResult, rec_y);
    static_assert(offsetof(SvtAmdLcuWork16, src_y) == offsetof(SvtAmdLcuWork, src_y) && offsetof(SvtAmdLcuResult16, rec_y) == offsetof(SvtAmdLcuResult, rec_y),
                  "the 8- and 16-bit records share their heads");
    if (!ctx || !works || !results || !lcus || !groups || !levels || !totals || n_lcus < 1 || work_stride < work_head || result_stride < result_head) {
        svt_amd_set_error("svt_amd_coeff_scan_picture: bad parameter");
        return SVT_AMD_ERR_BAD_PARAM;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = rate_tables_once(ctx->device);
    if (rc)
        return rc;
    const size_t n = (size_t)n_lcus;
    const size_t b_lcus = (n * sizeof(SvtAmdCoeffScanLcu) + 255) & ~(size_t)255, b_gpool = n * CS_MAX_SUBS * sizeof(SvtAmdCoeffScanGroup),
                 b_lpool = n * CS_MAX_LEVELS * sizeof(uint16_t);
    const size_t b_works = device_arrays ? 0 : ((n * work_head + 255) & ~(size_t)255), b_results = device_arrays ? 0 : ((n * result_head + 255) & ~(size_t)255);
    uint8_t *d = nullptr;
    rc = svt_amd_ctx_scratch(ctx, 256 + b_lcus + 2 * (b_gpool + b_lpool) + b_works + b_results, &d);
    if (rc)
        return rc;
    uint32_t *d_totals = (uint32_t *)d;
    SvtAmdCoeffScanLcu *d_lcus = (SvtAmdCoeffScanLcu *)(d + 256);
    SvtAmdCoeffScanGroup *d_gpool = (SvtAmdCoeffScanGroup *)(d + 256 + b_lcus), *d_groups = (SvtAmdCoeffScanGroup *)((uint8_t *)d_gpool + b_gpool);
    uint16_t *d_lpool = (uint16_t *)((uint8_t *)d_groups + b_gpool), *d_levels = (uint16_t *)((uint8_t *)d_lpool + b_lpool);
    const uint8_t *d_works = (const uint8_t *)works, *d_results = (const uint8_t *)results;
    if (!device_arrays) { /* only the heads travel: unit lists without the source samples, flags + coefficients without the reconstruction */
        uint8_t *dw = (uint8_t *)d_levels + b_lpool, *dr = dw + b_works;
        HIP_TRY(hipMemcpy2DAsync(dw, work_head, works, work_stride, work_head, n, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpy2DAsync(dr, result_head, results, result_stride, result_head, n, hipMemcpyHostToDevice, ctx->stream));
        d_works = dw, d_results = dr, work_stride = work_head, result_stride = result_head;
    }
    hipLaunchKernelGGL(k_coeff_scan, dim3((unsigned)n_lcus), dim3(256), 0, ctx->stream, d_works, work_stride, d_results, result_stride, d_lcus, d_gpool, d_lpool);
    hipLaunchKernelGGL(k_coeff_scan_bases, dim3(1), dim3(1024), 0, ctx->stream, d_lcus, n_lcus, d_totals);
    hipLaunchKernelGGL(k_coeff_scan_compact, dim3((unsigned)n_lcus), dim3(256), 0, ctx->stream, d_lcus, d_gpool, d_lpool, d_groups, group_capacity, d_levels,
                       level_capacity);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(totals, d_totals, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(lcus, d_lcus, n * sizeof(SvtAmdCoeffScanLcu), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (totals[0] > group_capacity || totals[1] > level_capacity) {
        svt_amd_set_error("svt_amd_coeff_scan_picture: %u groups / %u levels do not fit the capacities %u / %u", totals[0], totals[1], group_capacity, level_capacity);
        return SVT_AMD_ERR_RESOURCES;
    }
    if (totals[0])
        HIP_TRY(hipMemcpyAsync(groups, d_groups, (size_t)totals[0] * sizeof(SvtAmdCoeffScanGroup), hipMemcpyDeviceToHost, ctx->stream));
    if (totals[1])
        HIP_TRY(hipMemcpyAsync(levels, d_levels, (size_t)totals[1] * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_AMD_OK;
}
