"""Shared test/bench plumbing: ctypes views of include/svt_hevc_amd.h, loaders for
the three libraries (product HIP library, oracle restatement, reference build)
and the seeded synthetic clips of SURVEY.md section 8d.

Only tests/, bench.py and __graft_entry__.py import this module; the oracle and
reference loaders are test infrastructure (checker / CPU baseline only).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT_SO = os.path.join(ROOT, "svt-hevc_amd", "libsvt_hevc_amd.so")
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libsvtref.so")
REF_APP = os.path.join(ROOT, "oracle", "_ref", "SvtHevcEncApp_ref")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

ME_PU_COUNT = 85
PAD_FULL, PAD_QUARTER, PAD_SIXTEENTH = 68, 32, 16


class MeParams(C.Structure):
    """SvtAmdMeParams"""
    _fields_ = [
        ("luma_width", C.c_uint16), ("luma_height", C.c_uint16),
        ("num_lists", C.c_uint8), ("temporal_layer_index", C.c_uint8),
        ("ref_pocs_equal", C.c_uint8), ("enable_hme_flag", C.c_uint8),
        ("enable_hme_level0", C.c_uint8), ("enable_hme_level1", C.c_uint8),
        ("enable_hme_level2", C.c_uint8), ("one_quadrant_hme", C.c_uint8),
        ("update_hme_search_center", C.c_uint8), ("num_hme_regions_w", C.c_uint8),
        ("num_hme_regions_h", C.c_uint8), ("search_area_width", C.c_uint8),
        ("search_area_height", C.c_uint8), ("fractional_search_method", C.c_uint8),
        ("fractional_search_model", C.c_uint8), ("fractional_search_64x64", C.c_uint8),
        ("cu8x8_mode", C.c_uint8), ("cu16x16_mode", C.c_uint8),
        ("hme_l0_total_w", C.c_uint16), ("hme_l0_total_h", C.c_uint16),
        ("hme_l0_w", C.c_uint16 * 2), ("hme_l0_h", C.c_uint16 * 2),
        ("hme_l1_w", C.c_uint16 * 2), ("hme_l1_h", C.c_uint16 * 2),
        ("hme_l2_w", C.c_uint16 * 2), ("hme_l2_h", C.c_uint16 * 2),
        ("hme_l0_mult_x", C.c_uint16), ("hme_l0_mult_y", C.c_uint16),
        ("lambda_", C.c_uint32), ("mvd_bits", C.c_uint32 * 12),
    ]


class MeJob(C.Structure):
    """SvtAmdMeJob"""
    _fields_ = [("params", MeParams), ("cur_slot", C.c_int32), ("ref_slot", C.c_int32 * 2)]


class MeCuResult(C.Structure):
    """SvtAmdMeCuResult"""
    _fields_ = [
        ("x_mv_l0", C.c_int16), ("y_mv_l0", C.c_int16),
        ("x_mv_l1", C.c_int16), ("y_mv_l1", C.c_int16),
        ("distortion", C.c_uint32 * 3), ("direction", C.c_uint8 * 3),
        ("total_me_candidate_index", C.c_uint8),
    ]


class MeLcuResult(C.Structure):
    """SvtAmdMeLcuResult"""
    _fields_ = [
        ("pu", MeCuResult * ME_PU_COUNT),
        ("best_sad", (C.c_uint32 * ME_PU_COUNT) * 2),
        ("best_mv", (C.c_uint32 * ME_PU_COUNT) * 2),
        ("hme_center_x", C.c_int16 * 2), ("hme_center_y", C.c_int16 * 2),
        ("search_origin_x", C.c_int16 * 2), ("search_origin_y", C.c_int16 * 2),
        ("search_w", C.c_uint8 * 2), ("search_h", C.c_uint8 * 2),
    ]


# numpy dtypes with the same layout (checked in tests/test_abi.py)
ME_CU_DTYPE = np.dtype([("mv", "<i2", (4,)), ("distortion", "<u4", (3,)),
                        ("direction", "u1", (3,)), ("total", "u1")], align=True)
ME_LCU_DTYPE = np.dtype([("pu", ME_CU_DTYPE, (ME_PU_COUNT,)),
                         ("best_sad", "<u4", (2, ME_PU_COUNT)),
                         ("best_mv", "<u4", (2, ME_PU_COUNT)),
                         ("hme_center_x", "<i2", (2,)), ("hme_center_y", "<i2", (2,)),
                         ("search_origin_x", "<i2", (2,)), ("search_origin_y", "<i2", (2,)),
                         ("search_w", "u1", (2,)), ("search_h", "u1", (2,))], align=True)
ME_PARAMS_DTYPE = np.dtype(MeParams)

DUMP_DTYPE = np.dtype([("magic", "<u4"), ("record_size", "<u4"), ("picture_number", "<u8"),
                       ("ref_poc", "<u8", (2,)), ("lcu_index", "<u4"), ("lcu_origin_x", "<u4"),
                       ("lcu_origin_y", "<u4"), ("slice_type", "<u4"), ("enc_mode", "<u4"),
                       ("luma_crc", "<u4"), ("ref_crc", "<u4", (2,)),
                       ("params", ME_PARAMS_DTYPE), ("result", ME_LCU_DTYPE)], align=True)


class OisParams(C.Structure):
    """SvtAmdOisParams"""
    _fields_ = [("luma_width", C.c_uint16), ("luma_height", C.c_uint16), ("slice_is_intra", C.c_uint8),
                ("temporal_layer_index", C.c_uint8), ("limit_ois_to_dc_mode", C.c_uint8), ("skip_ois_8x8", C.c_uint8),
                ("cu8x8_mode", C.c_uint8), ("ois_kernel_level", C.c_uint8), ("ois_th_set", C.c_uint8),
                ("set_best_ois_distortion_to_valid", C.c_uint8)]


class OisJob(C.Structure):
    """SvtAmdOisJob"""
    _fields_ = [("params", OisParams), ("cur_slot", C.c_int32)]


class FrontendJob(C.Structure):
    """SvtAmdFrontendJob"""
    _fields_ = [("cur_slot", C.c_int32), ("ref_slot", C.c_int32 * 2), ("has_me", C.c_uint8), ("has_ois", C.c_uint8),
                ("compact", C.c_uint8), ("pad", C.c_uint8), ("me", MeParams), ("ois", OisParams)]


OIS_PARAMS_DTYPE = np.dtype(OisParams)
OIS_MAX_CAND = 18
OIS_LCU_DTYPE = np.dtype([("candidate", "<u4", (ME_PU_COUNT, OIS_MAX_CAND)), ("total", "u1", (ME_PU_COUNT,)),
                          ("pad", "u1", (3,))])
OIS_DUMP_DTYPE = np.dtype([("magic", "<u4"), ("record_size", "<u4"), ("picture_number", "<u8"), ("lcu_index", "<u4"),
                           ("slice_type", "<u4"), ("enc_mode", "<u4"), ("luma_crc", "<u4"),
                           ("params", OIS_PARAMS_DTYPE), ("me_sad", "<u4", (ME_PU_COUNT,)),
                           ("before", OIS_LCU_DTYPE), ("after", OIS_LCU_DTYPE)], align=True)
OIS_W_DIST, OIS_W_VALID, OIS_W_MODE = 1 << 21, 1 << 22, 1 << 23


def ois_params_from_record(rec):
    p = OisParams()
    C.memmove(C.byref(p), rec.tobytes(), C.sizeof(p))
    return p


def ois_apply(before, out):
    """What the reference's arrays hold after a call that produced `out` on top of `before` (OIS_LCU_DTYPE arrays):
    only the bitfields flagged SVT_AMD_OIS_W_* are replaced; total 0xFF = untouched."""
    c = out["candidate"].astype(np.uint32)
    m = (np.where(c & OIS_W_DIST, 0xFFFFF, 0) | np.where(c & OIS_W_VALID, 1 << 20, 0) |
         np.where(c & OIS_W_MODE, 0xFF000000, 0)).astype(np.uint32)
    after = before.copy()
    after["candidate"] = (before["candidate"] & ~m) | (c & m)
    after["total"] = np.where(out["total"] == 0xFF, before["total"], out["total"])
    return after


def oracle_ois_picture(oracle, params, luma, me_results=None):
    """Oracle OIS of every LCU of a picture. luma: HxW uint8; me_results: ME_LCU_DTYPE array or None."""
    h, w = luma.shape
    oracle.svt_oracle_ois_lcu.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_void_p]
    oracle.svt_oracle_ois_lcu.restype = None
    lw, lh = (w + 63) // 64, (h + 63) // 64
    out = np.zeros(lw * lh, OIS_LCU_DTYPE)
    luma = np.ascontiguousarray(luma)
    for l in range(lw * lh):
        sad = None
        if me_results is not None:
            sad = np.ascontiguousarray(me_results[l]["pu"]["distortion"][:, 0].astype(np.uint32))
        oracle.svt_oracle_ois_lcu(C.byref(params), luma.ctypes.data, w, (l % lw) * 64, (l // lw) * 64,
                                  sad.ctypes.data if sad is not None else None, out[l:l + 1].ctypes.data)
    return out


def params_from_record(rec):
    """numpy record with ME_PARAMS_DTYPE -> MeParams ctypes struct"""
    p = MeParams()
    C.memmove(C.byref(p), rec.tobytes(), C.sizeof(p))
    return p


# ----------------------------------------------------------------------------
# library loaders
# ----------------------------------------------------------------------------

_u8p = C.POINTER(C.c_uint8)


def _sig(fn, res, args):
    fn.restype = res
    fn.argtypes = args
    return fn


def _declare_leaf(lib, prefix):
    """Leaf-kernel prototypes shared by oracle (svt_oracle_) and product (svt_amd_)."""
    u32, u64, i16, vp = C.c_uint32, C.c_uint64, C.c_int16, C.c_void_p
    g = lambda n: getattr(lib, prefix + n)
    _sig(g("NxMSadKernel"), u32, [vp, u32, vp, u32, u32, u32])
    _sig(g("SadLoopKernel"), None, [vp, u32, vp, u32, u32, u32, C.POINTER(u64), C.POINTER(i16),
                                   C.POINTER(i16), u32, i16, i16])
    _sig(g("NxMSadAveragingKernel"), u32, [vp, u32, vp, u32, vp, u32, u32, u32])
    _sig(g("GetEightHorizontalSearchPointResults_8x8_16x16_PU"), None,
         [vp, u32, vp, u32, vp, vp, vp, vp, u32, vp])
    _sig(g("GetEightHorizontalSearchPointResults_32x32_64x64"), None, [vp, vp, vp, vp, vp, u32])
    _sig(g("SadCalculation_8x8_16x16"), None, [vp, u32, vp, u32, vp, vp, vp, vp, u32, vp])
    _sig(g("SadCalculation_32x32_64x64"), None, [vp, vp, vp, vp, vp, u32])
    _sig(g("AvcStyleLumaInterpolationFilterHorizontal"), None, [vp, u32, vp, u32, u32, u32, vp, u32])
    _sig(g("AvcStyleLumaInterpolationFilterVertical"), None, [vp, u32, vp, u32, u32, u32, vp, u32])
    _sig(g("PictureAverageKernel"), None, [vp, u32, vp, u32, vp, u32, u32, u32])
    _sig(g("SpatialFullDistortionKernel"), u64, [vp, u32, vp, u32, u32, u32])
    _sig(g("Decimation2D"), None, [vp, u32, u32, u32, vp, u32, u32])


def load_oracle():
    """Test infrastructure: the CPU restatement (oracle/liboracle.so)."""
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    lib = C.CDLL(os.environ.get("SVT_ORACLE_LIB") or ORACLE_SO)   # SVT_ORACLE_LIB: an instrumented build of the same sources (tools/md_logic_coverage.sh)
    _declare_leaf(lib, "svt_oracle_")
    _sig(lib.svt_oracle_picture_create, C.c_void_p, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32])
    _sig(lib.svt_oracle_picture_destroy, None, [C.c_void_p])
    _sig(lib.svt_oracle_me_picture, C.c_int,
         [C.POINTER(MeParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p])
    return lib


def load_ref():
    """Test infrastructure: the reference itself (oracle/_ref/libsvtref.so), or None."""
    if not os.path.exists(REF_SO):
        return None
    return C.CDLL(REF_SO)


def load_product():
    """The HIP library.  Fails loudly when it has not been built."""
    if not os.path.exists(PRODUCT_SO):
        raise RuntimeError("svt-hevc_amd/libsvt_hevc_amd.so missing: run `python __graft_entry__.py build`")
    # torch bundles its own copy of the HIP runtime with the same SONAME (libamdhip64.so.7).  Import it
    # FIRST so the loader binds our library to that copy: two HIP/HSA runtimes in one process cannot both
    # open the GPU ("No HIP GPUs are available" from whichever initialises second).
    import torch  # noqa: F401
    lib = C.CDLL(os.environ.get("SVT_PRODUCT_LIB", PRODUCT_SO))  # the override: A/B runs of an experimental build (tools/)
    _declare_leaf(lib, "svt_amd_")
    vp, i, u16, u32 = C.c_void_p, C.c_int, C.c_uint16, C.c_uint32
    _sig(lib.svt_amd_context_create, i, [i, u16, u16, i, C.POINTER(vp)])
    _sig(lib.svt_amd_context_destroy, None, [vp])
    _sig(lib.svt_amd_version, C.c_char_p, [])
    _sig(lib.svt_amd_last_error, C.c_char_p, [])
    _sig(lib.svt_amd_picture_upload, i, [vp, i, vp, u32, u16, u16])
    _sig(lib.svt_amd_picture_upload_device, i, [vp, i, vp, u32, u16, u16])
    _sig(lib.svt_amd_me_picture, i, [vp, C.POINTER(MeParams), i, C.POINTER(C.c_int), vp])
    _sig(lib.svt_amd_me_picture_launch, i, [vp, C.POINTER(MeParams), i, C.POINTER(C.c_int)])
    _sig(lib.svt_amd_me_picture_fetch, i, [vp, i, vp])
    _sig(lib.svt_amd_me_batch_launch, i, [vp, C.POINTER(MeJob), i])
    _sig(lib.svt_amd_me_picture_range_launch, i, [vp, C.POINTER(MeParams), i, C.POINTER(C.c_int), u32, u32])
    _sig(lib.svt_amd_ois_picture, i, [vp, C.POINTER(OisParams), i, vp, vp])
    _sig(lib.svt_amd_ois_picture_launch, i, [vp, C.POINTER(OisParams), i])
    _sig(lib.svt_amd_ois_picture_fetch, i, [vp, i, vp])
    _sig(lib.svt_amd_ois_batch_launch, i, [vp, C.POINTER(OisJob), i])
    _sig(lib.svt_amd_synchronize, i, [vp])
    _sig(lib.svt_amd_timer_begin, i, [vp])
    _sig(lib.svt_amd_timer_end, i, [vp, C.POINTER(C.c_float)])
    _sig(lib.svt_amd_kernel_time, i, [vp, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int)])
    _sig(lib.svt_amd_picture_read_plane, i, [vp, i, i, vp, C.c_size_t, C.POINTER(u32),
                                             C.POINTER(u32), C.POINTER(u32)])
    # lanes + asynchronous front-end pipeline
    _sig(lib.svt_amd_context_fork, i, [vp, C.POINTER(vp)])
    _sig(lib.svt_amd_picture_upload_async, i, [vp, i, vp, u32, u16, u16])
    _sig(lib.svt_amd_picture_publish, i, [vp, i])
    _sig(lib.svt_amd_frontend_submit, i, [vp, C.POINTER(FrontendJob)])
    _sig(lib.svt_amd_frontend_wait, i, [vp, C.POINTER(vp), C.POINTER(vp)])
    _sig(lib.svt_amd_frontend_release, i, [vp])
    _sig(lib.svt_amd_frontend_warmup, i, [vp])
    _sig(lib.svt_amd_lane_event_record, i, [vp, i])
    _sig(lib.svt_amd_records_pack_batch_async, i, [vp, C.POINTER(C.c_int), i, i, vp, vp])
    _sig(lib.svt_amd_lane_event_wait, i, [vp, vp, i])
    _sig(lib.svt_amd_ois_compact_candidates, i, [C.POINTER(OisParams)])
    _sig(lib.svt_amd_me_picture_fetch_compact_async, i, [vp, i, vp])
    _sig(lib.svt_amd_ois_picture_fetch_compact_async, i, [vp, i, i, vp])
    _sig(lib.svt_amd_device_alloc, i, [vp, C.c_size_t, C.POINTER(vp)])
    _sig(lib.svt_amd_device_free, i, [vp, vp])
    _sig(lib.svt_amd_device_upload, i, [vp, vp, vp, C.c_size_t])
    _sig(lib.svt_amd_device_download, i, [vp, vp, vp, C.c_size_t])
    _sig(lib.svt_amd_device_upload_async, i, [vp, vp, vp, C.c_size_t])
    _sig(lib.svt_amd_device_download_async, i, [vp, vp, vp, C.c_size_t])
    _sig(lib.svt_amd_host_alloc, i, [vp, C.c_size_t, C.POINTER(vp)])
    _sig(lib.svt_amd_host_free, i, [vp, vp])
    _sig(lib.svt_amd_me_picture_fetch_async, i, [vp, i, vp])
    _sig(lib.svt_amd_ois_picture_fetch_async, i, [vp, i, vp])
    _sig(lib.svt_amd_picture_upload_device_batch, i, [vp, i, C.POINTER(C.c_int), C.POINTER(vp), u32, u16, u16])
    return lib


# ----------------------------------------------------------------------------
# synthetic clips
# ----------------------------------------------------------------------------

def gen_luma(kind, w, h, t, seed):
    """One 8-bit luma frame of clip `kind` at time t."""
    if kind == "flat":
        return np.full((h, w), 128, np.uint8)
    if kind == "static":
        # the first frame of "motion" standing still under mild per-frame noise: zero vectors win and the zero-vector merge candidates carry real residual
        frame_rng = np.random.default_rng(seed * 1000 + 77 + t)
        return np.clip(gen_luma("motion", w, h, 0, seed).astype(np.int32) + frame_rng.integers(-2, 3, size=(h, w)), 0, 255).astype(np.uint8)
    rng = np.random.default_rng(seed)
    if kind == "noise":
        for _ in range(t):
            rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        return rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    if kind == "objects":
        # textured rectangles moving at different (fractional) speeds over a textured background that itself pans slowly, plus mild per-frame
        # noise: neighbouring units get different motion vectors, uncovered areas have no match (AMVP, uni-prediction and intra candidates win)
        tex = rng.integers(0, 256, size=(h + 64, w + 64), dtype=np.uint8)
        x = np.arange(w)[None, :]
        y = np.arange(h)[:, None]
        bx, by = (t * 3) // 4, t // 2
        l = 96 + 40 * np.sin((x + bx) / 29.0) * np.cos((y + by) / 37.0) + (tex[by:by + h, bx:bx + w] >> 3).astype(np.float64)
        nobj = 6 + (w * h) // (320 * 240)
        for k in range(nobj):
            ow, oh = int(rng.integers(24, 97)), int(rng.integers(24, 97))
            px, py = float(rng.integers(0, w)), float(rng.integers(0, h))
            vx, vy = float(rng.integers(-14, 15)) / 4.0, float(rng.integers(-10, 11)) / 4.0
            base, amp = int(rng.integers(40, 200)), int(rng.integers(2, 6))
            otex = rng.integers(0, 256, size=(oh, ow), dtype=np.uint8) >> amp
            x0, y0 = int(np.floor(px + vx * t)) % w, int(np.floor(py + vy * t)) % h
            x1, y1 = min(w, x0 + ow), min(h, y0 + oh)
            l[y0:y1, x0:x1] = base + otex[:y1 - y0, :x1 - x0]
        frame_rng = np.random.default_rng(seed * 1000 + t)
        l = l + frame_rng.integers(-3, 4, size=(h, w))
        return np.clip(np.floor(l), 0, 255).astype(np.uint8)
    # "motion": sinusoids drifting 3/2 px per frame + fixed noise field moving (-2,-1)
    noise = rng.integers(0, 256, size=(h + 256, w + 512), dtype=np.uint8)
    x = np.arange(w)[None, :]
    y = np.arange(h)[:, None]
    tt = t % 128
    l = 128 + 50 * (np.sin((x + 3 * t) / 17.0) + np.cos((y + 2 * t) / 23.0)) + \
        (noise[tt:tt + h, 2 * tt:2 * tt + w] >> 4)
    return np.clip(np.floor(l), 0, 255).astype(np.uint8)


def gen_chroma(w, h, t):
    xc = np.arange(w // 2)[None, :]
    yc = np.arange(h // 2)[:, None]
    cb = 128 + 40 * np.sin((xc / 2.0 + t) / 31.0) + 0 * yc
    cr = 128 + 40 * np.cos((yc / 2.0 - t) / 29.0) + 0 * xc
    return (np.clip(np.floor(cb), 0, 255).astype(np.uint8),
            np.clip(np.floor(cr), 0, 255).astype(np.uint8))


def write_clip(path, kind, w, h, n, seed):
    with open(path, "wb") as f:
        for t in range(n):
            f.write(gen_luma(kind, w, h, t, seed).tobytes())
            cb, cr = gen_chroma(w, h, t)
            f.write(cb.tobytes())
            f.write(cr.tobytes())


def write_clip10(path, kind, w, h, n, seed):
    """10-bit clip in the reference app's unpacked format (16-bit little-endian samples): (8-bit clip << 2) | 2 noise bits."""
    rng = np.random.default_rng(seed + 1000)
    with open(path, "wb") as f:
        for t in range(n):
            cb, cr = gen_chroma(w, h, t)
            for plane in (gen_luma(kind, w, h, t, seed), cb, cr):
                lsb = rng.integers(0, 4, plane.shape, dtype=np.uint16)
                f.write(((plane.astype(np.uint16) << 2) | lsb).astype("<u2").tobytes())


def write_clip10_compressed(path, kind, w, h, n, seed):
    """10-bit clip in the reference's compressed format (-compressed-ten-bit-format 1; Docs/svt-hevc_encoder_user_guide.md
    "Compressed 10-bit format", read by Source/App/EbAppProcessCmd.c:881-900): per frame the three 8-bit MSB planes, then the
    three 2-bit planes packed four samples to a byte (luma W*H/4 bytes in the 64x64-unrolled order, then Cb, Cr).  Every byte
    value is a legal quadruple of 2-bit samples, so the LSB planes are seeded random bytes."""
    rng = np.random.default_rng(seed + 2000)
    with open(path, "wb") as f:
        for t in range(n):
            cb, cr = gen_chroma(w, h, t)
            for plane in (gen_luma(kind, w, h, t, seed), cb, cr):
                f.write(plane.tobytes())
            for size in (w * h // 4, w * h // 16, w * h // 16):
                f.write(rng.integers(0, 256, size, dtype=np.uint8).tobytes())


# ---- encode-pass contracts (SvtAmdLcuWork / SvtAmdLcuResult, include/svt_hevc_amd.h) ----
LCU_CU_DTYPE = np.dtype([("x", "u1"), ("y", "u1"), ("size", "u1"), ("pred_mode", "u1"), ("intra_luma_mode", "u1"), ("bottom_left_ok", "u1"),
                         ("top_right_ok", "u1"), ("qp", "u1"), ("chroma_qp", "u1"), ("leaf_index", "u1"), ("inter_dir", "u1"), ("inter_kind", "u1"), ("dz_offset", "<u4"),
                         ("mv", "<i2", (2, 2))])
LCU_WORK_DTYPE = np.dtype([("lcu_x", "<u2"), ("lcu_y", "<u2"), ("num_cus", "u1"), ("slice_type", "u1"), ("temporal_layer", "u1"),
                           ("constrained_intra", "u1"), ("strong_smoothing", "u1"), ("tile_left", "u1"), ("tile_top", "u1"), ("tile_right", "u1"),
                           ("pad", "u1", 4), ("full_lambda", "<u4"), ("luma_cbf_bits", "<u4", 4), ("pm_core", "u1"), ("pad2", "u1", 11), ("cu", LCU_CU_DTYPE, 64),
                           ("src_y", "u1", 4096), ("src_cb", "u1", 1024), ("src_cr", "u1", 1024)])
LCU_CU_RESULT_DTYPE = np.dtype([("cbf", "u1", 3), ("only_dc", "u1", 3), ("nz", "<u2", 3)])
LCU_RESULT_DTYPE = np.dtype([("cu", LCU_CU_RESULT_DTYPE, 64), ("coeff_y", "<i2", 4096), ("coeff_cb", "<i2", 1024), ("coeff_cr", "<i2", 1024),
                             ("rec_y", "u1", 4096), ("rec_cb", "u1", 1024), ("rec_cr", "u1", 1024)])
LCU_BORDER_DTYPE = np.dtype([("lcu_x", "<u2"), ("lcu_y", "<u2"), ("mode_bottom", "u1", 16), ("mode_right", "u1", 16), ("bottom_y", "u1", 64),
                             ("right_y", "u1", 64), ("bottom_cb", "u1", 32), ("right_cb", "u1", 32), ("bottom_cr", "u1", 32), ("right_cr", "u1", 32)])
EP_RECORD_DTYPE = np.dtype([("magic", "<u4"), ("record_size", "<u4"), ("picture_number", "<u8"), ("width", "<u4"), ("height", "<u4"),
                            ("lcu_index", "<u4"), ("dlf_off", "<u4"), ("ref_poc", "<u8", 2), ("work", LCU_WORK_DTYPE), ("result", LCU_RESULT_DTYPE)])
# ---- device-resident mode decision (include/svt_hevc_amd.h "Device-resident mode decision") ----
CABAC_COST_BYTES = 4 * (60 * 2 + 28 * 2) + 2 * 42 + 2 * 24 + 2 * 6 + 2 * 4 + 2 * (24 // 4 * 16) + 32 * 16
MD_RATES_DTYPE = np.dtype([("splitFlagBits", "<u4", 6), ("skipFlagBits", "<u4", 6), ("mvpIndexBits", "<u4", 2), ("intraPartSizeBits", "<u4", 2),
                           ("interPartSizeBits", "<u4", 8), ("predModeBits", "<u4", 2), ("intraLumaBits", "<u4", 4), ("intraChromaBits", "<u4", 5),
                           ("refPicBits", "<u4", 3), ("mvdBits", "<u4", 12), ("lumaCbfBits", "<u4", 10), ("chromaCbfBits", "<u4", 10),
                           ("rootCbfBits", "<u4", 2), ("transSubDivFlagBits", "<u4", 6), ("mergeFlagBits", "<u4", 2), ("mergeIndexBits", "<u4", 5),
                           ("saoMergeFlagBits", "<u4", 2), ("saoTypeIndexBits", "<u4", 6), ("saoOffsetTrunUnaryBits", "<u4", 8),
                           ("interBiDirBits", "<u4", 8), ("interUniDirBits", "<u4", 2), ("pad", "<u4", 17)])
assert MD_RATES_DTYPE.itemsize == 512
MD_PICTURE_DTYPE = np.dtype([("width", "<u2"), ("height", "<u2")] + [(n, "u1") for n in (
    "slice_type", "temporal_layer", "is_reference", "enc_mode", "depth_mode", "intra_md_open_loop", "intra_injection_method", "limit_intra",
    "mpm_search", "mpm_search_candidate", "pf_md_level", "nfl_level_md", "nmm_level_md", "full_loop_escape", "single_fast_loop",
    "coeff_cabac_update", "spatial_sse_full_loop", "chroma_level", "intra4x4_level", "rdoq_pmcore_method", "skip_ois_8x8", "cu8x8_mode",
    "cu16x16_mode", "limit_ois_to_dc_mode", "constrained_intra", "strong_smoothing", "qp", "chroma_qp", "intra8x8_restriction_inter_slice")] +
    [("pad", "u1", 3)] +
    [(n, "<u4") for n in ("fast_lambda", "full_lambda", "fast_chroma_lambda", "full_chroma_lambda")] + [("rates", MD_RATES_DTYPE)])
MD_LCU_DTYPE = np.dtype([("leaf_count", "u1"), ("leaf_index", "u1", 85), ("leaf_split", "u1", 85), ("tile_left", "u1"), ("tile_top", "u1"),
                         ("tile_right", "u1"), ("is_complete", "u1"), ("complexity_status_2", "u1"), ("contouring_class", "u1", 4),
                         ("chroma_encode_mode", "u1"), ("restrict_intra_global_motion", "u1"), ("lcu_md_mode", "u1"), ("skip_small_cu", "u1"),
                         ("cmplx_noise", "u1"), ("variance_below_200", "u1"), ("edge_block", "u1"), ("no_stop_split", "u1"), ("pad", "u1", 3)])
MD_LCU_OUT_DTYPE = np.dtype([("split", "u1", 85), ("tested", "u1", 85), ("pred_mode", "u1", 85), ("intra_luma_mode", "u1", 85), ("ycbf", "u1", 85),
                             ("inter_dir", "u1", 85), ("merge_flag", "u1", 85), ("merge_index", "u1", 85), ("pad", "u1", 8), ("mv", "<i2", (85, 2, 2)),
                             ("cost", "<u8", 85), ("merge_cost", "<u8", 85), ("skip_cost", "<u8", 85)])
MD_TMVP_LCU_DTYPE = np.dtype([("mv", "<i2", (2, 16, 2)), ("ref_poc", "<u8", (2, 16)), ("pred_dir", "u1", 16), ("available", "u1", 16)])
MD_INTER_DTYPE = np.dtype([("picture_number", "<u8"), ("ref_poc", "<u8", 2), ("colocated_poc", "<u8")] + [(n, "u1") for n in (
    "colocated_pu_ref_list", "is_low_delay", "tmvp_enable", "use_subpel", "unrestricted_mv", "generate_amvp_table_md", "extra_injection",
    "improve_sharpness", "skip_cost_bias")] + [("pad", "u1", 3), ("chroma_weight", "<u4")])
assert MD_LCU_DTYPE.itemsize == 191 and MD_LCU_OUT_DTYPE.itemsize == 3408 and MD_TMVP_LCU_DTYPE.itemsize == 416 and MD_INTER_DTYPE.itemsize == 48
MD_PIC_MAGIC, MD_LCU_MAGIC = 0x4350444D, 0x434C444D
MD_PIC_RECORD_DTYPE = np.dtype([("magic", "<u4"), ("record_size", "<u4"), ("picture_number", "<u8"), ("nlcu", "<u4"), ("has_inter", "<u4"),
                                ("pic", MD_PICTURE_DTYPE), ("cost", "u1", CABAC_COST_BYTES), ("pad2", "u1", 4), ("inter", MD_INTER_DTYPE)] +
                               [(n, "<u4") for n in ("ref_stride_y", "ref_stride_c", "ref_origin_x", "ref_origin_y", "ref_width", "ref_height", "nref",
                                                     "tmvp_present")])
assert MD_PIC_RECORD_DTYPE.itemsize == 2232 and MD_PICTURE_DTYPE.itemsize == 564
MD_LCU_RECORD_DTYPE = np.dtype([("magic", "<u4"), ("record_size", "<u4"), ("picture_number", "<u8"), ("lcu_index", "<u4"), ("pad", "<u4"),
                                ("lcu", MD_LCU_DTYPE), ("pad1", "u1"), ("out", MD_LCU_OUT_DTYPE)])
assert MD_LCU_RECORD_DTYPE.itemsize == 3624 and MD_LCU_RECORD_DTYPE.fields["out"][1] == 216
LCU_WORK16_DTYPE = np.dtype([(n, LCU_WORK_DTYPE.fields[n][0]) if not n.startswith("src_") else (n, "<u2", LCU_WORK_DTYPE.fields[n][0].shape)
                             for n in LCU_WORK_DTYPE.names])
LCU_RESULT16_DTYPE = np.dtype([(n, LCU_RESULT_DTYPE.fields[n][0]) if not n.startswith("rec_") else (n, "<u2", LCU_RESULT_DTYPE.fields[n][0].shape)
                               for n in LCU_RESULT_DTYPE.names])
EP_RECORD16_DTYPE = np.dtype([("magic", "<u4"), ("record_size", "<u4"), ("picture_number", "<u8"), ("width", "<u4"), ("height", "<u4"),
                              ("lcu_index", "<u4"), ("dlf_off", "<u4"), ("ref_poc", "<u8", 2), ("work", LCU_WORK16_DTYPE), ("result", LCU_RESULT16_DTYPE)])
# picture-level records of the encode-pass dump (oracle/ref_harness_encodepass_dump.c)
EP_REF_HEAD_DTYPE = np.dtype([("magic", "<u4"), ("record_size", "<u4"), ("poc", "<u8"), ("bps", "<u4"), ("strideY", "<u4"), ("strideC", "<u4"),
                              ("originX", "<u4"), ("originY", "<u4"), ("width", "<u4"), ("height", "<u4"), ("pad", "<u4")])
EP_MAGIC, EP_REF_MAGIC, EP_COST_MAGIC = 0x53415045, 0x46525045, 0x43435045


class RefPicture(C.Structure):
    """SvtAmdRefPicture (include/svt_hevc_amd.h)"""
    _fields_ = [("d_y", C.c_void_p), ("d_cb", C.c_void_p), ("d_cr", C.c_void_p), ("strideY", C.c_uint32), ("strideC", C.c_uint32),
                ("originX", C.c_uint32), ("originY", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32)]


INTER_AMVP, INTER_MERGE, INTER_SKIP = 0, 1, 2
LCU_BORDER16_DTYPE = np.dtype([("lcu_x", "<u2"), ("lcu_y", "<u2"), ("mode_bottom", "u1", 16), ("mode_right", "u1", 16), ("bottom_y", "<u2", 64),
                               ("right_y", "<u2", 64), ("bottom_cb", "<u2", 32), ("right_cb", "<u2", 32), ("bottom_cr", "<u2", 32), ("right_cr", "<u2", 32)])


def plane_checksum(luma):
    """Same polynomial as oracle/ref_harness_me_dump.c:plane_checksum (s = s*31 + v mod 2^32)."""
    flat = luma.astype(np.uint64).ravel()
    # Horner in blocks to stay vectorised
    s = 0
    p31 = 31
    chunk = 4096
    pw = np.ones(chunk, np.uint64)
    for i in range(chunk - 2, -1, -1):
        pw[i] = (pw[i + 1] * p31) & 0xFFFFFFFF
    for off in range(0, flat.size, chunk):
        blk = flat[off:off + chunk]
        k = blk.size
        acc = int((blk * pw[chunk - k:] & 0xFFFFFFFF).sum() & 0xFFFFFFFF)
        s = (s * pow(31, k, 1 << 32) + acc) & 0xFFFFFFFF
    return s


# ----------------------------------------------------------------------------
# oracle convenience wrappers
# ----------------------------------------------------------------------------

class OraclePicture:
    def __init__(self, lib, luma):
        self.lib = lib
        luma = np.ascontiguousarray(luma, np.uint8)
        h, w = luma.shape
        self.handle = lib.svt_oracle_picture_create(luma.ctypes.data, w, w, h)
        if not self.handle:
            raise MemoryError("svt_oracle_picture_create")

    def close(self):
        if self.handle:
            self.lib.svt_oracle_picture_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


def lcu_count(w, h):
    return ((w + 63) // 64) * ((h + 63) // 64)


def oracle_me_picture(lib, params, cur, ref0, ref1=None, lcu_begin=0, lcu_end=None):
    n = lcu_count(params.luma_width, params.luma_height)
    if lcu_end is None:
        lcu_end = n
    out = np.zeros(n, ME_LCU_DTYPE)
    rc = lib.svt_oracle_me_picture(C.byref(params), cur.handle, ref0.handle,
                                   ref1.handle if ref1 is not None else None,
                                   lcu_begin, lcu_end, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("svt_oracle_me_picture -> %d" % rc)
    return out


def compare_me(got, want, num_lists, what="", lcus=None):
    """Bit-exact comparison of the fields the reference defines.  For P pictures the
    reference leaves xMvL1/yMvL1 uninitialised (EbMotionEstimation.c:4427-4434)."""
    errs = []
    idx = range(len(want)) if lcus is None else lcus
    for i in idx:
        g, w = got[i], want[i]
        for pu in range(ME_PU_COUNT):
            gp, wp = g["pu"][pu], w["pu"][pu]
            tot = int(wp["total"])
            if int(gp["total"]) != tot:
                errs.append((i, pu, "total", int(gp["total"]), tot))
                continue
            nmv = 4 if num_lists == 2 else 2
            if not np.array_equal(gp["mv"][:nmv], wp["mv"][:nmv]):
                errs.append((i, pu, "mv", gp["mv"][:nmv].tolist(), wp["mv"][:nmv].tolist()))
            if not np.array_equal(gp["distortion"][:tot], wp["distortion"][:tot]):
                errs.append((i, pu, "dist", gp["distortion"][:tot].tolist(), wp["distortion"][:tot].tolist()))
            if not np.array_equal(gp["direction"][:tot], wp["direction"][:tot]):
                errs.append((i, pu, "dir", gp["direction"][:tot].tolist(), wp["direction"][:tot].tolist()))
        for l in range(num_lists):
            if not np.array_equal(g["best_sad"][l], w["best_sad"][l]):
                errs.append((i, -1, "best_sad[%d]" % l, None, None))
            if not np.array_equal(g["best_mv"][l], w["best_mv"][l]):
                errs.append((i, -1, "best_mv[%d]" % l, None, None))
            if int(g["search_origin_x"][l]) != int(w["search_origin_x"][l]) or \
               int(g["search_origin_y"][l]) != int(w["search_origin_y"][l]):
                errs.append((i, -1, "origin[%d]" % l,
                             (int(g["search_origin_x"][l]), int(g["search_origin_y"][l])),
                             (int(w["search_origin_x"][l]), int(w["search_origin_y"][l]))))
    assert not errs, "%s: %d mismatches, first: %s" % (what, len(errs), errs[:5])


# ---- entropy hand-off pre-scan (include/svt_hevc_amd.h: SvtAmdCoeffScanTu / Group / Lcu) ----
COEFF_SCAN_TU_DTYPE = np.dtype([("scan_index", "u1"), ("last_scan_set", "i1"), ("pos_last", "u1"), ("last_x", "u1"), ("last_y", "u1"), ("dc_only", "u1"),
                                ("first_group", "<u2")])
COEFF_SCAN_GROUP_DTYPE = np.dtype([("sigmap", "<u2"), ("sign", "<u2"), ("gt1", "<u2"), ("first_level", "<u2")])
COEFF_SCAN_LCU_DTYPE = np.dtype([("group_base", "<u4"), ("level_base", "<u4"), ("groups", "<u2"), ("levels", "<u2"), ("pad", "u1", 4),
                                 ("tu", COEFF_SCAN_TU_DTYPE, (3, 64))])
assert COEFF_SCAN_TU_DTYPE.itemsize == 8 and COEFF_SCAN_GROUP_DTYPE.itemsize == 8 and COEFF_SCAN_LCU_DTYPE.itemsize == 16 + 3 * 64 * 8


# ---- picture-analysis statistics: SvtAmdPaLcuStats, and the record of oracle/ref_harness_me_dump.c (SVT_REF_PA_DUMP) ----
PA_LCU_STATS_DTYPE = np.dtype([("variance", "<u2", 85), ("y_mean", "u1", 85), ("pad", "u1")])
PA_DUMP_DTYPE = np.dtype([("magic", "<u4"), ("kind", "<u4"), ("picture_number", "<u8"), ("lcu_index", "<u4"), ("regions_w", "<u4"), ("regions_h", "<u4"),
                          ("pad", "<u4"), ("variance", "<u2", 85), ("y_mean", "u1", 85), ("average_intensity", "u1"), ("histogram", "<u4", (4, 4, 256)),
                          ("region_average", "u1", (4, 4))], align=True)
assert PA_LCU_STATS_DTYPE.itemsize == 256
