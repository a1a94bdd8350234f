"""-m gpu: parity of the HIP motion-estimation front half, through the C-ABI, against
(1) the committed golden fixtures recorded from the reference, (2) the oracle on seeded
clips with varied controls, and (3) size-independent properties at BASELINE sizes."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from golden_util import golden_cases, kept_lcus, load_case
from gpu_util import default_params, me_picture, read_plane, upload

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,w,h", [("motion", 640, 384), ("noise", 328, 264), ("motion", 1920, 1080)])
def test_prep_planes_match_oracle(product, gpu_ctx, oracle, kind, w, h):
    """pad / decimate / half-pel planes vs the oracle's picture construction."""
    luma = S.gen_luma(kind, w, h, 3, 7)
    upload(product, gpu_ctx, 0, luma)
    pic = S.OraclePicture(oracle, luma)

    class Plane(C.Structure):
        _fields_ = [("data", C.c_void_p), ("stride", C.c_uint32), ("pad", C.c_uint32),
                    ("width", C.c_uint32), ("height", C.c_uint32)]

    planes = (Plane * 6).from_address(pic.handle)
    for which, name in enumerate(["full", "quarter", "sixteenth", "hp_b", "hp_h", "hp_j"]):
        pl = planes[which]
        rows = pl.height + 2 * pl.pad
        want = np.ctypeslib.as_array((C.c_uint8 * (rows * pl.stride)).from_address(pl.data)).reshape(rows, pl.stride)
        got = read_plane(product, gpu_ctx, 0, which, w, h)
        assert got.shape == want.shape, name
        m = 0 if which < 3 else 2  # half-pel planes: the oracle leaves a 2-sample frame unwritten
        sl = (slice(m, rows - m), slice(m, pl.stride - m))
        assert np.array_equal(got[sl], want[sl]), "plane %s differs" % name


@pytest.mark.parametrize("name", golden_cases())
def test_me_matches_reference_golden(product, gpu_ctx, name):
    """HIP ME == what the reference's MotionEstimateLcu produced in a real encoder run."""
    g = load_case(name)
    slots = {}
    own_ctx = None
    if g["w"] > 1920 or g["h"] > 1088:  # the 4K fixture (BASELINE config 3) needs its own, larger context
        own_ctx = C.c_void_p()
        assert product.svt_amd_context_create(0, g["w"], (g["h"] + 7) & ~7, 3, C.byref(own_ctx)) == 0, product.svt_amd_last_error()
        gpu_ctx = own_ctx

    def slot(n):
        if n not in slots:
            assert len(slots) < (3 if own_ctx else 6)
            slots[n] = len(slots)
            upload(product, gpu_ctx, slots[n], S.gen_luma(g["kind"], g["w"], g["h"], n, g["seed"]))
        return slots[n]

    for i, (pn, slice_type, r0, r1) in enumerate(g["meta"]):
        p = S.params_from_record(g["params"][i])
        refs = [slot(int(r0))] + ([slot(int(r1))] if p.num_lists == 2 else [])
        got = me_picture(product, gpu_ctx, p, slot(int(pn)), refs)
        S.compare_me(got, g["results"][i], p.num_lists, "%s picture %d" % (name, pn), kept_lcus(g))
    if own_ctx:
        product.svt_amd_context_destroy(own_ctx)


VARIANTS = [
    dict(),
    dict(num_lists=2, temporal_layer_index=1, cu8x8_mode=0),
    dict(num_lists=2, temporal_layer_index=2, enable_hme_level2=1, ref_pocs_equal=1),
    dict(fractional_search_model=0, fractional_search_method=1, cu8x8_mode=0),
    dict(fractional_search_model=0, fractional_search_method=2, fractional_search_64x64=1, cu8x8_mode=0),
    dict(fractional_search_model=2, enable_hme_flag=0, update_hme_search_center=0),
    dict(search_area_width=21, search_area_height=13, enable_hme_level1=0),
    dict(search_area_width=75, search_area_height=70, temporal_layer_index=3),
    dict(one_quadrant_hme=1, enable_hme_level1=0, hme_l0_total_w=64, hme_l0_total_h=32),
    dict(num_lists=2, cu16x16_mode=1, hme_l0_mult_x=140, hme_l0_mult_y=70, temporal_layer_index=1),
]


@pytest.mark.parametrize("vi", range(len(VARIANTS)))
@pytest.mark.parametrize("kind,w,h", [("motion", 448, 328), ("noise", 256, 192), ("flat", 192, 128)])
def test_me_matches_oracle(product, gpu_ctx, oracle, kind, w, h, vi):
    """Seeded clips incl. ragged sizes (w,h not multiples of 64) and all-ties input."""
    frames = [S.gen_luma(kind, w, h, t, 11 + vi) for t in range(3)]
    for s, f in enumerate(frames):
        upload(product, gpu_ctx, s, f)
    pics = [S.OraclePicture(oracle, f) for f in frames]
    p = default_params(w, h, **VARIANTS[vi])
    got = me_picture(product, gpu_ctx, p, 1, [0, 2])
    want = S.oracle_me_picture(oracle, p, pics[1], pics[0], pics[2] if p.num_lists == 2 else None)
    S.compare_me(got, want, p.num_lists, "%s %dx%d variant %d" % (kind, w, h, vi))
    for k in ("hme_center_x", "hme_center_y", "search_w", "search_h"):
        assert np.array_equal(got[k][:, :p.num_lists], want[k][:, :p.num_lists]), k


def test_me_8k_m4_rows_match_oracle(product, oracle):
    """BASELINE configs[4] size through the C-ABI: a 7680x4320 B picture (8,160 LCUs, ~150 MB per device picture slot) with the
    encMode-4 controls of SURVEY Appendix B (SSD sub-pel search on all 85 PUs, 8x8 refinement, HME L0 64x32 + L1, full-pel 16x9:
    the 4K-class table row the reference uses for 8K, taken from the 4K fixture and switched to SSD / model 0 / cu8x8Mode 0).
    The whole picture runs on the device; the oracle (pinned on the same paths by the encMode-4 fixture) checks four LCU rows:
    top, two interior, and the partial bottom row."""
    w, h = 7680, 4320
    g = load_case("b_3840x2160_m7")
    p = S.params_from_record(g["params"][0])
    p.luma_width, p.luma_height = w, h
    p.fractional_search_method, p.fractional_search_model, p.cu8x8_mode = 2, 0, 0
    ctx = C.c_void_p()
    assert product.svt_amd_context_create(0, w, h, 3, C.byref(ctx)) == 0, product.svt_amd_last_error()
    try:
        frames = [S.gen_luma("motion", w, h, t, 7) for t in range(3)]
        for s_, f in enumerate(frames):
            upload(product, ctx, s_, f)
        got = me_picture(product, ctx, p, 1, [0, 2])
        pics = [S.OraclePicture(oracle, f) for f in frames]
        wl = (w + 63) // 64
        for row in (0, 29, 47, 67):
            want = S.oracle_me_picture(oracle, p, pics[1], pics[0], pics[2], row * wl, (row + 1) * wl)
            S.compare_me(got, want, 2, "8K row %d" % row, range(row * wl, (row + 1) * wl))
    finally:
        product.svt_amd_context_destroy(ctx)


@pytest.mark.parametrize("case,bands", [("b_1024x768_m7", (0, 3, 4, 9, 12)), ("p_1920x1080_m9", (0, 1, 8, 16, 17))])
def test_me_row_bands_equal_the_full_picture(product, case, bands):
    """svt_amd_me_picture_range_launch (the multi-GPU cut of one picture by LCU rows): the records of every band, taken
    right after that band's launch, are together byte-identical to the one-launch picture - B pictures included, where list 1 reads
    list 0's result of the same LCU."""
    g = load_case(case)
    p = S.params_from_record(g["params"][0])
    w, h = p.luma_width, p.luma_height
    wl, hl = (w + 63) // 64, (h + 63) // 64
    ctx = C.c_void_p()
    assert product.svt_amd_context_create(0, w, (h + 7) & ~7, 3, C.byref(ctx)) == 0, product.svt_amd_last_error()
    try:
        for s_ in range(3):
            upload(product, ctx, s_, S.gen_luma("motion", w, h, s_, 11))
        full = me_picture(product, ctx, p, 1, [0, 2])
        refs = (C.c_int * 2)(0, 2)
        nl = wl * hl
        poison = np.full(nl, 0xA5, np.uint8).repeat(S.ME_LCU_DTYPE.itemsize).view(S.ME_LCU_DTYPE)
        rows = sorted(set(r for r in bands if r < hl) | {0, hl})
        got = poison.copy()
        for a, b in zip(rows[:-1], rows[1:]):
            assert product.svt_amd_me_picture_range_launch(ctx, C.byref(p), 1, refs, a * wl, b * wl) == 0, product.svt_amd_last_error()
            band = np.zeros(nl, S.ME_LCU_DTYPE)
            assert product.svt_amd_me_picture_fetch(ctx, 1, band.ctypes.data) == 0, product.svt_amd_last_error()
            got[a * wl:b * wl] = band[a * wl:b * wl]
        assert got.tobytes() == full.tobytes()
    finally:
        product.svt_amd_context_destroy(ctx)


def test_me_full_size_properties(product, gpu_ctx):
    """BASELINE config 2 size (1920x1080): properties that need no CPU oracle."""
    w, h = 1920, 1080
    f0 = S.gen_luma("motion", w, h, 0, 7)
    upload(product, gpu_ctx, 0, f0)
    upload(product, gpu_ctx, 1, f0)
    p = default_params(w, h)
    # identical pictures: every PU must find SAD 0 at MV (0,0)
    got = me_picture(product, gpu_ctx, p, 0, [1])
    assert (got["pu"]["distortion"][:, :, 0] == 0).all()
    # (the partial bottom LCU row also matches one row up inside the replicated padding,
    #  which raster order visits first - so only complete LCU rows must return (0,0))
    assert (got["pu"]["mv"][:16 * 30, :, :2] == 0).all()
    # pure translation by (-8,+4): interior LCUs recover the exact vector with zero SAD
    f1 = np.roll(f0, (4, -8), axis=(0, 1))
    upload(product, gpu_ctx, 2, f1)
    got = me_picture(product, gpu_ctx, p, 2, [0])
    wl = 30
    inner = [r * wl + c for r in range(2, 14) for c in range(2, 28)]
    assert (got["pu"]["distortion"][inner, :, 0] == 0).all()
    assert (got["pu"]["mv"][inner, :, 0] == 8 * 4).all() and (got["pu"]["mv"][inner, :, 1] == -4 * 4).all()
    # run-to-run determinism
    again = me_picture(product, gpu_ctx, p, 2, [0])
    assert got.tobytes() == again.tobytes()
    # SAD tree consistency: best 64x64 SAD >= sum of best 32x32 SADs >= sum of best 16x16 ...
    bs = got["best_sad"][:, 0, :].astype(np.int64)
    assert (bs[:, 0] >= bs[:, 1:5].sum(1)).all() and (bs[:, 1:5].sum(1) >= bs[:, 5:21].sum(1)).all()


def test_bad_arguments_are_rejected(product, gpu_ctx):
    p = default_params(640, 384)
    refs = (C.c_int * 2)(0, 0)
    out = np.zeros(60, S.ME_LCU_DTYPE)
    assert product.svt_amd_me_picture(gpu_ctx, C.byref(p), 99, refs, out.ctypes.data) == -1
    upload(product, gpu_ctx, 0, S.gen_luma("flat", 320, 256, 0, 1))
    assert product.svt_amd_me_picture(gpu_ctx, C.byref(p), 0, refs, out.ctypes.data) == -1  # size mismatch
    p.num_lists = 3
    assert product.svt_amd_me_picture(gpu_ctx, C.byref(p), 0, refs, out.ctypes.data) == -1
    small = np.zeros((32, 32), np.uint8)
    assert product.svt_amd_picture_upload(gpu_ctx, 0, small.ctypes.data, 32, 32, 32) == -1


def test_batched_prep_equals_single(product, gpu_ctx):
    """svt_amd_picture_upload_device_batch (one fused launch for several pictures) builds the same six planes as
    the per-picture upload, including odd source strides (slow path of the kernel)."""
    import torch
    w, h = 328, 264
    frames = [S.gen_luma("motion", w, h, t, 5) for t in range(3)]
    for stride in (w, w + 3):
        buf = np.zeros((3, h, stride), np.uint8)
        for i, f in enumerate(frames):
            buf[i, :, :w] = f
        dev = torch.from_numpy(buf).cuda()
        slots = (C.c_int * 3)(3, 4, 5)
        ptrs = (C.c_void_p * 3)(*[dev[i].data_ptr() for i in range(3)])
        product.svt_amd_picture_upload_device_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p),
                                                                C.c_uint32, C.c_uint16, C.c_uint16]
        torch.cuda.synchronize()  # torch fills/copies run on torch's stream, the library on its own
        assert product.svt_amd_picture_upload_device_batch(gpu_ctx, 3, slots, ptrs, stride, w, h) == 0, \
            product.svt_amd_last_error()
        for i, f in enumerate(frames):
            upload(product, gpu_ctx, 0, f)
            for which in range(6):
                a = read_plane(product, gpu_ctx, 0, which, w, h)
                b = read_plane(product, gpu_ctx, 3 + i, which, w, h)
                assert np.array_equal(a, b), (stride, i, which)
