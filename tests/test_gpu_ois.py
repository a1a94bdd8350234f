"""-m gpu: open-loop intra search on the device, through the C-ABI, against (1) the committed before/after dumps of
the reference's OpenLoopIntraSearchLcu, (2) the oracle on seeded clips with varied controls (incl. the carried
bestMode corner), (3) ME -> OIS chained on the device at BASELINE size."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from gpu_util import default_params, me_picture, upload
from test_oracle_ois_golden import CASES, kept_lcus, load_ois_case, me_like

pytestmark = pytest.mark.gpu


def ois_picture(lib, ctx, params, slot, me=None):
    n = S.lcu_count(params.luma_width, params.luma_height)
    out = np.zeros(n, S.OIS_LCU_DTYPE)
    rc = lib.svt_amd_ois_picture(ctx, C.byref(params), slot, me.ctypes.data if me is not None else None, out.ctypes.data)
    assert rc == 0, lib.svt_amd_last_error()
    return out


def same(a, b):
    return np.array_equal(a["candidate"], b["candidate"]) and np.array_equal(a["total"], b["total"])


def test_struct_sizes():
    assert C.sizeof(S.OisParams) == 12 and S.OIS_LCU_DTYPE.itemsize == 85 * 18 * 4 + 88


@pytest.mark.parametrize("name", CASES)
def test_ois_matches_reference_golden(product, gpu_ctx, name):
    g, kind, w, h, seed = load_ois_case(name)
    ctx = gpu_ctx
    if w > 1920:  # BASELINE configs[2] size: its own context
        ctx = C.c_void_p()
        assert product.svt_amd_context_create(0, w, (h + 7) & ~7, 1, C.byref(ctx)) == 0, product.svt_amd_last_error()
    try:
        kept = kept_lcus(g, w, h)
        for i, (pn, slice_type, enc_mode) in enumerate(g["meta"]):
            upload(product, ctx, 0, S.gen_luma(kind, w, h, int(pn), seed))
            params = S.ois_params_from_record(g["params"][i])
            me = me_like(g["me_sad"][i], kept, S.lcu_count(w, h)) if slice_type != 2 else None
            out = ois_picture(product, ctx, params, 0, me)[kept]
            got = S.ois_apply(g["before"][i], out)
            assert same(got, g["after"][i]), (name, int(pn))
    finally:
        if ctx is not gpu_ctx:
            product.svt_amd_context_destroy(ctx)


def mk_params(w, h, **kw):
    p = S.OisParams()
    p.luma_width, p.luma_height = w, h
    p.ois_th_set = 1
    for k, v in kw.items():
        setattr(p, k, v)
    return p


VARIANTS = [
    dict(slice_is_intra=1),
    dict(slice_is_intra=1, skip_ois_8x8=1, cu8x8_mode=1),          # I slices ignore both
    dict(skip_ois_8x8=1, cu8x8_mode=1),
    dict(skip_ois_8x8=0, cu8x8_mode=0, ois_th_set=2, temporal_layer_index=2),
    dict(skip_ois_8x8=0, cu8x8_mode=1, ois_th_set=0, temporal_layer_index=3, set_best_ois_distortion_to_valid=1),
    dict(limit_ois_to_dc_mode=1, skip_ois_8x8=1),
    dict(ois_kernel_level=1, skip_ois_8x8=0, cu8x8_mode=0),
    dict(ois_kernel_level=1, skip_ois_8x8=1),
]


@pytest.mark.parametrize("kind,w,h", [("motion", 328, 264), ("noise", 320, 256), ("flat", 192, 136), ("blocks", 256, 256)])
@pytest.mark.parametrize("v", range(len(VARIANTS)))
def test_ois_matches_oracle_variants(product, gpu_ctx, oracle, kind, w, h, v):
    if kind == "blocks":
        # white picture, a black 32x32 CU at (32,32) of every LCU whose left / top / bottom-left neighbours are all
        # white: H, V and mode 2 all give SAD == 32*32*255, so no mode beats the initial best and the reference
        # keeps bestMode (and stage1SadArray) from the previous CU - the carried-state path of the kernel.
        # Noise in the first 28 columns of each LCU gives the earlier CUs varied best modes.
        luma = np.full((h, w), 255, np.uint8)
        rng0 = np.random.default_rng(3)
        for lx in range(0, w, 64):
            luma[:, lx:lx + 28] = rng0.integers(0, 256, (h, min(28, w - lx)), dtype=np.uint8)
            for ly in range(0, h, 64):
                luma[ly + 32:ly + 64, lx + 32:lx + 64] = 0
    else:
        luma = S.gen_luma(kind, w, h, 2, 9)
    upload(product, gpu_ctx, 0, luma)
    params = mk_params(w, h, **VARIANTS[v])
    rng = np.random.default_rng(v)
    me = None
    if not params.slice_is_intra:
        n = S.lcu_count(w, h)
        # ME SADs spread around the DC SADs so every OIS point is hit
        me = me_like(rng.integers(0, 6000, (n, 85)).astype(np.uint32) * (rng.integers(0, 4, (n, 85)) > 0))
    want = S.oracle_ois_picture(oracle, params, luma, me)
    got = ois_picture(product, gpu_ctx, params, 0, me)
    assert same(got, want)
    if not params.slice_is_intra and not params.limit_ois_to_dc_mode and not params.ois_kernel_level and kind == "motion":
        assert len(set(want["total"][:, 1:21].ravel().tolist()) - {255}) >= 3  # several OIS points exercised


def test_me_then_ois_on_device_1080p(product, gpu_ctx, oracle):
    """BASELINE cfg2 shape: ME results stay on the device and feed OIS (me = NULL)."""
    w, h = 1920, 1080
    f0, f1 = S.gen_luma("motion", w, h, 0, 7), S.gen_luma("motion", w, h, 1, 7)
    upload(product, gpu_ctx, 0, f0)
    upload(product, gpu_ctx, 1, f1)
    mp = default_params(w, h)
    me = me_picture(product, gpu_ctx, mp, 1, [0])
    params = mk_params(w, h, skip_ois_8x8=1, cu8x8_mode=1)
    got = ois_picture(product, gpu_ctx, params, 1, None)
    want = S.oracle_ois_picture(oracle, params, f1, me)
    assert same(got, want)
    assert (want["total"][:, 1:21] != 255).all(axis=1)[:30 * 16].all()  # complete LCU rows: every 32/16 CU decided


def test_ois_batch_matches_single(product, gpu_ctx, oracle):
    """Several pictures in one launch (grid = pictures x LCUs) == one launch per picture."""
    w, h = 416, 240
    mp = default_params(w, h)
    frames = [S.gen_luma("motion", w, h, t, 7) for t in range(4)]
    for t, f in enumerate(frames):
        upload(product, gpu_ctx, t, f)
    mes = {t: me_picture(product, gpu_ctx, mp, t, [t - 1]) for t in (1, 2, 3)}
    params = mk_params(w, h, skip_ois_8x8=1, cu8x8_mode=1)
    jobs = (S.OisJob * 3)()
    for i, t in enumerate((1, 2, 3)):
        jobs[i].params, jobs[i].cur_slot = params, t
    assert product.svt_amd_ois_batch_launch(gpu_ctx, jobs, 3) == 0, product.svt_amd_last_error()
    n = S.lcu_count(w, h)
    for t in (1, 2, 3):
        got = np.zeros(n, S.OIS_LCU_DTYPE)
        assert product.svt_amd_ois_picture_fetch(gpu_ctx, t, got.ctypes.data) == 0
        assert same(got, S.oracle_ois_picture(oracle, params, frames[t], mes[t]))


@pytest.mark.parametrize("kind,w,h", [("motion", 416, 240), ("flat", 320, 256), ("noise", 328, 264), ("motion", 1920, 1080)])
def test_zz_sad_matches_oracle(product, gpu_ctx, oracle, kind, w, h):
    """ComputeDecimatedZzSad on the device (1/16 planes of two slots) vs the oracle."""
    from test_oracle_zz import ZZ, oracle_zz
    cur, prev = S.gen_luma(kind, w, h, 3, 7), S.gen_luma(kind, w, h, 2, 7)
    if kind == "motion":
        cur = cur.copy()
        cur[:, :192] = prev[:, :192]
    upload(product, gpu_ctx, 0, prev)
    upload(product, gpu_ctx, 1, cur)
    got = np.zeros(S.lcu_count(w, h), ZZ)
    product.svt_amd_zz_sad_picture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    assert product.svt_amd_zz_sad_picture(gpu_ctx, 1, 0, got.ctypes.data) == 0, product.svt_amd_last_error()
    want = oracle_zz(oracle, cur, prev)
    assert np.array_equal(got, want)
