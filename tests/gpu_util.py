"""Helpers for the -m gpu tests: drive the HIP library through its C-ABI."""
import ctypes as C

import numpy as np

import svtlib as S


def upload(lib, ctx, slot, luma):
    luma = np.ascontiguousarray(luma, np.uint8)
    h, w = luma.shape
    rc = lib.svt_amd_picture_upload(ctx, slot, luma.ctypes.data, w, w, h)
    assert rc == 0, lib.svt_amd_last_error()


def me_picture(lib, ctx, params, cur_slot, ref_slots):
    n = S.lcu_count(params.luma_width, params.luma_height)
    out = np.zeros(n, S.ME_LCU_DTYPE)
    refs = (C.c_int * 2)(ref_slots[0], ref_slots[1] if len(ref_slots) > 1 else ref_slots[0])
    rc = lib.svt_amd_me_picture(ctx, C.byref(params), cur_slot, refs, out.ctypes.data)
    assert rc == 0, lib.svt_amd_last_error()
    return out


def read_plane(lib, ctx, slot, which, w, h):
    pads = [68, 32, 16, 68, 68, 68]
    sh = [0, 1, 2, 0, 0, 0][which]
    pw, ph = (w >> sh) + 2 * pads[which], (h >> sh) + 2 * pads[which]
    buf = np.zeros((ph, pw), np.uint8)
    st, pad, rows = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = lib.svt_amd_picture_read_plane(ctx, slot, which, buf.ctypes.data, buf.size, C.byref(st), C.byref(pad),
                                        C.byref(rows))
    assert rc == 0, lib.svt_amd_last_error()
    assert (st.value, pad.value, rows.value) == (pw, pads[which], ph)
    return buf


def default_params(w, h, **kw):
    """cfg2-like controls (1080p encMode 9 LDP as dumped from the reference), overridable."""
    p = S.MeParams()
    p.luma_width, p.luma_height = w, h
    p.num_lists = 1
    p.temporal_layer_index = 0
    p.enable_hme_flag = p.enable_hme_level0 = p.enable_hme_level1 = 1
    p.enable_hme_level2 = 0
    p.update_hme_search_center = 1
    p.num_hme_regions_w = p.num_hme_regions_h = 2
    p.search_area_width, p.search_area_height = 8, 7
    p.fractional_search_method = 0
    p.fractional_search_model = 1
    p.cu8x8_mode = 1
    p.hme_l0_total_w, p.hme_l0_total_h = 48, 40
    for k in range(2):
        p.hme_l0_w[k], p.hme_l0_h[k] = 24, 20
        p.hme_l1_w[k], p.hme_l1_h[k] = 4, 4
        p.hme_l2_w[k], p.hme_l2_h[k] = 4, 4
    p.hme_l0_mult_x = p.hme_l0_mult_y = 100
    p.lambda_ = 203
    for k, v in enumerate([16384, 49152, 20480, 40960, 20480, 40960, 24576, 36864, 24576, 36864, 0, 0]):
        p.mvd_bits[k] = v
    for k, v in kw.items():
        setattr(p, k, v)
    return p
