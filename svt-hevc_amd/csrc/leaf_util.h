/* Host-pointer staging helpers shared by the LEAF-layer wrappers (leaf.hip, leaf_txfm.hip). */
#ifndef SVT_AMD_LEAF_UTIL_H
#define SVT_AMD_LEAF_UTIL_H
#include <stdio.h>
#include "svt_amd_internal.h"

/* -------- device buffer helper -------- */
struct DBuf {
    uint8_t *d = nullptr;
    size_t n = 0;
    bool ok = true;
    DBuf(const void *host, size_t bytes, bool upload = true) : n(bytes)
    {
        if (hipMalloc((void **)&d, bytes + 64) != hipSuccess) {
            ok = false;
            d = nullptr;
            svt_amd_set_error("leaf: hipMalloc(%zu) failed", bytes);
            return;
        }
        if (upload && host && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
            ok = false;
            svt_amd_set_error("leaf: H2D copy failed");
        }
    }
    bool download(void *host, size_t bytes) const
    {
        if (!ok || hipMemcpy(host, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
            svt_amd_set_error("leaf: D2H copy failed");
            return false;
        }
        return true;
    }
    ~DBuf()
    {
        if (d)
            (void)hipFree(d);
    }
};
static inline size_t span(uint32_t stride, uint32_t w, uint32_t h) { return h ? (size_t)(h - 1) * stride + w : 0; }
static inline bool finish(const char *what)
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess)
        e = hipGetLastError();
    if (e != hipSuccess) {
        svt_amd_set_error("leaf %s: %s", what, hipGetErrorString(e));
        fprintf(stderr, "svt_hevc_amd: %s\n", svt_amd_last_error());
        return false;
    }
    return true;
}


#endif
