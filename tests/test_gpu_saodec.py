"""-m gpu: the SAO parameter decision of a whole picture (svt_amd_sao_decide_picture) through the C-ABI against (1) whole
pictures of real encoder runs (tests/golden/saodec_*.npz: the statistics the reference gathered per LCU in, the parameters and
costs it decided out) and (2) the oracle (pinned to the same pictures in tests/test_oracle_saodec_golden.py) on random
statistics: 8 / 10 bit, full and reduced modes, every temporal layer, shut-off LCUs, tile edges, up to 300 LCU rows."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_saodec_golden import CASES, DEC, LCU, STATS, oracle_decide_picture, pictures_of, same_decision

pytestmark = pytest.mark.gpu
vp, u32 = C.c_void_p, C.c_uint32


def gpu_decide_picture(product, gpu_ctx, pic):
    import torch
    n = pic["cols"] * pic["rows"]
    dev = [torch.from_numpy(np.ascontiguousarray(pic["stats"][c]).view(np.uint8)).cuda() for c in range(3)]
    en = torch.from_numpy(pic["enable"]).cuda() if pic["enable"] is not None else None
    params = torch.from_numpy(pic["params"].copy().view(np.uint8)).cuda()
    costs = torch.full((n, 2), -1, dtype=torch.int64).cuda()
    product.svt_amd_sao_decide_picture.argtypes = [vp, vp, vp, vp, vp, u32, u32, vp, vp, vp]
    torch.cuda.synchronize()
    rc = product.svt_amd_sao_decide_picture(gpu_ctx, pic["P"].ctypes.data, dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(),
                                            pic["cols"], pic["rows"], en.data_ptr() if en is not None else None, params.data_ptr(),
                                            costs.data_ptr())
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    return params.cpu().numpy().view(LCU), costs.cpu().numpy()


@pytest.mark.parametrize("name", CASES)
def test_sao_decision_matches_reference_pictures(product, gpu_ctx, name):
    for k, pic in enumerate(pictures_of(name)):
        params, costs = gpu_decide_picture(product, gpu_ctx, pic)
        for i in pic["idx"]:
            assert same_decision(params[i], pic["want"][i]), (name, k, int(i), params[i], pic["want"][i])
        assert (costs[pic["idx"]] == pic["costs"][pic["idx"]]).all()
        off = pic["enable"] == 0
        assert (params["type"][off] == 0).all() and (params["merge_left"][off] == 0).all() and (params["merge_up"][off] == 0).all()


@pytest.mark.parametrize("name", CASES)
def test_sao_decision_lcu_form_matches_reference_records(product, gpu_ctx, name):
    """the one-LCU host form: every record's neighbours are handed in exactly as the reference saw them"""
    from test_oracle_saodec_golden import load_saodec_case, lcu_of, params_of, stats_of
    recs = load_saodec_case(name)
    recs = recs[:: max(1, len(recs) // 150)]
    product.svt_amd_sao_decide_lcu.argtypes = [vp] * 9
    for i, r in enumerate(recs):
        st, d, out, costs = stats_of(r), params_of(r), np.zeros(1, LCU), np.zeros(2, np.int64)
        left, up = lcu_of(r["left"]), lcu_of(r["up"])
        rc = product.svt_amd_sao_decide_lcu(gpu_ctx, d.ctypes.data, st[0:1].ctypes.data, st[1:2].ctypes.data, st[2:3].ctypes.data,
                                            left.ctypes.data if r["has_left"] else None, up.ctypes.data if r["has_up"] else None,
                                            out.ctypes.data, costs.ctypes.data)
        assert rc == 0, product.svt_amd_last_error()
        assert same_decision(out[0], r["out"]), (name, i, out[0], r["out"])
        assert costs[0] == r["luma_cost"] and costs[1] == r["chroma_cost"], (name, i, costs)


def random_picture(rng, cols, rows, is10, mm, layer, with_enable, flat):
    n = cols * rows
    P = np.zeros(1, DEC)
    P["lambda"], P["chroma_lambda"] = int(rng.integers(1 << 17, 1 << 23)), int(rng.integers(1 << 17, 1 << 22))
    P["type_bits"] = rng.integers(20000, 120000, 6)
    P["merge_bits"] = rng.integers(5000, 90000, 2)
    P["offset_bits"] = np.sort(rng.integers(10000, 300000, 8))
    P["is_10bit"], P["mm_sao"], P["temporal_layer"] = is10, mm, layer
    stats = np.zeros((3, n), STATS)
    amp = 31 if is10 else 7
    for c in range(3):
        cnt = rng.integers(0, 400, (n, 32)) * (rng.random((n, 32)) < 0.7)
        stats[c]["boCount"] = cnt
        stats[c]["boDiff"] = (cnt * rng.integers(-amp - 3, amp + 4, (n, 32)) + rng.integers(-50, 51, (n, 32))) * (cnt > 0)
        cnt = rng.integers(0, 1500, (n, 4, 5)) * (rng.random((n, 4, 5)) < 0.9)
        sign = np.array([1, 1, -1, -1, 0]) if not flat else np.array([1, -1, 1, -1, 0])
        stats[c]["eoCount"] = cnt
        stats[c]["eoDiff"] = (cnt * rng.integers(0, amp + 3, (n, 4, 5)) * sign // (1 + 3 * flat) + rng.integers(-80, 81, (n, 4, 5))) * (cnt > 0)
    if flat:    # neighbouring LCUs alike: merging pays
        for c in range(3):
            for k in range(1, n):
                if rng.random() < 0.6:
                    stats[c][k] = stats[c][k - 1] if rng.random() < 0.5 or k < cols else stats[c][k - cols]
    params = np.zeros(n, LCU)
    params["edge_flags"] = rng.integers(0, 16, n) * (rng.random(n) < 0.15)
    enable = (rng.random(n) < 0.85).astype(np.uint8) if with_enable else None
    if with_enable:     # a few LCUs with parameters "decided by an earlier call"
        given = rng.random(n) < 0.08
        enable[given] = 2
        params["type"][given] = rng.integers(0, 6, (int(given.sum()), 2))
        params["type"][:, 1][params["type"][:, 1] == 5] = 3
        params["offset"][given] = rng.integers(0, amp + 1, (int(given.sum()), 3, 4)) * np.array([1, 1, -1, -1])
        params["band"][given] = rng.integers(0, 29, (int(given.sum()), 3))
    return dict(P=P, stats=stats, enable=enable, params=params, cols=cols, rows=rows)


@pytest.mark.parametrize("cols,rows,is10,mm,layer,with_enable,flat", [
    (1, 1, 0, 1, 0, 0, 0), (7, 4, 0, 1, 0, 1, 1), (7, 4, 1, 1, 1, 1, 1), (30, 17, 0, 1, 2, 1, 0), (30, 17, 0, 0, 0, 1, 1), (30, 17, 1, 0, 1, 0, 1),
    (30, 17, 0, 0, 2, 1, 0), (60, 34, 0, 1, 0, 1, 1), (60, 34, 1, 1, 3, 0, 1), (3, 300, 0, 1, 0, 1, 1), (120, 68, 0, 1, 1, 1, 1)])
def test_sao_decision_matches_oracle_random(product, gpu_ctx, oracle, cols, rows, is10, mm, layer, with_enable, flat):
    rng = np.random.default_rng(cols * 131 + rows * 7 + is10 * 3 + mm + layer * 17)
    pic = random_picture(rng, cols, rows, is10, mm, layer, with_enable, flat)
    if pic["enable"] is None:
        pic_o = dict(pic, enable=np.ones(cols * rows, np.uint8))
    else:
        pic_o = pic
    want, wcost = oracle_decide_picture(oracle, pic_o)
    got, gcost = gpu_decide_picture(product, gpu_ctx, pic)
    bad = [i for i in range(cols * rows) if got[i].tobytes() != want[i].tobytes()]
    assert not bad, (len(bad), bad[:5], got[bad[0]], want[bad[0]])
    decided = pic_o["enable"] != 2      # "given" LCUs: costs are left alone
    assert (gcost[decided] == wcost[decided]).all()
    if mm or layer < 2:
        assert (want["type"][:, 0] != 0).sum() > 0
        if cols * rows > 100 and flat:
            assert want["merge_left"].sum() > 0 and want["merge_up"].sum() > 0


def test_sao_decision_rejects_bad_arguments(product, gpu_ctx):
    product.svt_amd_sao_decide_picture.argtypes = [vp, vp, vp, vp, vp, u32, u32, vp, vp, vp]
    P = np.zeros(1, DEC)
    assert product.svt_amd_sao_decide_picture(gpu_ctx, P.ctypes.data, None, None, None, 1, 1, None, None, None) != 0
    assert product.svt_amd_sao_decide_picture(gpu_ctx, None, 8, 8, 8, 1, 1, None, 8, 8) != 0
    assert product.svt_amd_sao_decide_picture(gpu_ctx, P.ctypes.data, 8, 8, 8, 0, 1, None, 8, 8) != 0
