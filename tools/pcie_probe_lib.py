"""GPU box: PCIe rates of the library's own pinned buffers and stream copies (H2D one large copy, D2H many picture-sized copies)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import svtlib as S
lib = S.load_product()
vp = C.c_void_p
root, la, lb = vp(), vp(), vp()
assert lib.svt_amd_context_create(0, 640, 384, 1, C.byref(root)) == 0
lib.svt_amd_context_fork(root, C.byref(la)); lib.svt_amd_context_fork(root, C.byref(lb))
n = 512 << 20
h, h2, d, d2 = vp(), vp(), vp(), vp()
assert lib.svt_amd_host_alloc(root, n, C.byref(h)) == 0 and lib.svt_amd_host_alloc(root, n, C.byref(h2)) == 0
assert lib.svt_amd_device_alloc(root, n, C.byref(d)) == 0 and lib.svt_amd_device_alloc(root, n, C.byref(d2)) == 0
def t(fn, reps=6):
    fn(); lib.svt_amd_synchronize(la); lib.svt_amd_synchronize(lb)
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    lib.svt_amd_synchronize(la); lib.svt_amd_synchronize(lb)
    return (time.perf_counter() - t0) / reps
up = lambda: lib.svt_amd_device_upload_async(la, d, h, n)
def down(chunk):
    def f():
        for o in range(0, n, chunk):
            lib.svt_amd_device_download_async(lb, vp(h2.value + o), vp(d2.value + o), chunk)
    return f
print("H2D 512 MiB in one copy: %.1f GB/s" % (n / t(up) / 1e9))
for ch in (512 << 20, 8 << 20, 4 << 20, 1 << 20):
    print("D2H 512 MiB in %4d MiB copies: %.1f GB/s" % (ch >> 20, n / t(down(ch)) / 1e9))
both = lambda: (up(), down(4 << 20)())
print("both directions at once (4 MiB D2H copies): %.1f GB/s total" % (2 * n / t(both) / 1e9))
