// Probe: semantics of v_qsad_pk_u16_u8 on gfx950 (used by the HME / full-pel search kernels).
// expected: out[i] = acc[i] + sum_k |ref[i+k] - src[k]|, k = 0..3, i = 0..3, no saturation below 65535.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint64_t *a, const uint32_t *b, const uint64_t *c, uint64_t *o)
{
    o[threadIdx.x] = __builtin_amdgcn_qsad_pk_u16_u8(a[threadIdx.x], b[threadIdx.x], c[threadIdx.x]);
}
int main()
{
    const int N = 64;
    uint64_t ha[N], hc[N], ho[N];
    uint32_t hb[N];
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (int i = 0; i < N; i++) {
        ha[i] = ((uint64_t)rnd() << 32) | rnd();
        hb[i] = rnd();
        hc[i] = i < 32 ? 0 : (((uint64_t)rnd() << 32) | rnd());
        if (i == 63) ha[i] = 0, hb[i] = 0xffffffffu, hc[i] = 0xfff0fff0fff0fff0ull; // overflow behaviour
    }
    uint64_t *da, *dc, *dout; uint32_t *db;
    hipMalloc(&da, sizeof ha); hipMalloc(&dc, sizeof hc); hipMalloc(&dout, sizeof ho); hipMalloc(&db, sizeof hb);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(N), 0, 0, da, db, dc, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < N; i++) {
        uint64_t want = 0;
        for (int p = 0; p < 4; p++) {
            uint32_t sad = 0;
            for (int kk = 0; kk < 4; kk++) {
                int r = (int)((ha[i] >> (8 * (p + kk))) & 255), q = (int)((hb[i] >> (8 * kk)) & 255);
                sad += (uint32_t)(r > q ? r - q : q - r);
            }
            want |= (uint64_t)((sad + (uint32_t)((hc[i] >> (16 * p)) & 0xffff)) & 0xffff) << (16 * p);
        }
        if (want != ho[i]) { bad++; printf("lane %d: got %016llx want(wrap) %016llx\n", i, (unsigned long long)ho[i], (unsigned long long)want); }
    }
    printf("qsad probe: %d mismatches (lane 63 tests u16 overflow: wrap expected by this check)\n", bad);
    return 0;
}
