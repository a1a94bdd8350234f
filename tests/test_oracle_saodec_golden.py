"""CPU-only: pins oracle/svt_oracle_saodec.c:svt_oracle_sao_decide_lcu against records of real SaoGenerationDecision /
SaoGenerationDecision16bit calls of the reference's encode pass (tests/golden/saodec_*.npz, made by
tests/golden/make_saodec_golden.py): same statistics, lambdas, rate tables and neighbours in, same parameters and costs out."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import svtlib as S

CASES = sorted(os.path.basename(p)[7:-4] for p in glob.glob(os.path.join(S.GOLDEN_DIR, "saodec_*.npz")))
assert len(CASES) == 7
STATS = np.dtype([("boDiff", "<i4", 32), ("boCount", "<u2", 32), ("eoDiff", "<i4", (4, 5)), ("eoCount", "<u2", (4, 5))], align=True)
LCU = np.dtype([("merge_left", "u1"), ("merge_up", "u1"), ("edge_flags", "u1"), ("pad", "u1"), ("type", "<u4", 2),
                ("offset", "<i4", (3, 4)), ("band", "<u4", 3)])
DEC = np.dtype([("lambda", "<u8"), ("chroma_lambda", "<u8"), ("type_bits", "<u4", 6), ("merge_bits", "<u4", 2),
                ("offset_bits", "<u4", 8), ("is_10bit", "u1"), ("mm_sao", "u1"), ("temporal_layer", "u1"), ("pad", "u1")], align=True)


def load_saodec_case(name):
    return np.load(os.path.join(S.GOLDEN_DIR, "saodec_%s.npz" % name))["recs"]


def pictures_of(name):
    """the fixture's records regrouped into whole pictures: decision params, per-LCU statistics of the three components,
    enable map (0 = the encode pass never ran the decision there), edge flags in, the reference's parameters / costs"""
    g = np.load(os.path.join(S.GOLDEN_DIR, "saodec_%s.npz" % name))
    recs, w, h = g["recs"], int(g["width"]), int(g["height"])
    cols, rows = (w + 63) // 64, (h + 63) // 64
    out = []
    for pn in np.unique(recs["picture_number"]):
        rr = recs[recs["picture_number"] == pn]
        idx = (rr["origin_y"] // 64) * cols + rr["origin_x"] // 64
        assert len(np.unique(idx)) == len(idx)
        stats, enable = np.zeros((3, cols * rows), STATS), np.zeros(cols * rows, np.uint8)
        params, want, costs = np.zeros(cols * rows, LCU), np.zeros(cols * rows, LCU), np.zeros((cols * rows, 2), np.int64)
        enable[idx] = 1
        for c in range(3):
            stats[c]["boDiff"][idx], stats[c]["boCount"][idx] = rr["bo_diff"][:, c], rr["bo_count"][:, c]
            stats[c]["eoDiff"][idx], stats[c]["eoCount"][idx] = rr["eo_diff"][:, c], rr["eo_count"][:, c]
        # LCUs the decision skipped still are merge candidates (all-off parameters) unless a tile edge lies between
        params["edge_flags"] = 5
        params["edge_flags"][idx] = (rr["has_left"] == 0) * 1 + (rr["has_up"] == 0) * 4
        for k in ("merge_left", "merge_up", "type", "offset", "band"):
            want[k][idx] = rr["out"][k]
        costs[idx, 0], costs[idx, 1] = rr["luma_cost"], rr["chroma_cost"]
        out.append(dict(P=params_of(rr[0]), stats=stats, enable=enable, params=params, want=want, costs=costs, cols=cols, rows=rows,
                        idx=idx))
    return out


def oracle_decide_picture(oracle, pic):
    params, costs = pic["params"].copy(), np.zeros((pic["cols"] * pic["rows"], 2), np.int64)
    st = [np.ascontiguousarray(pic["stats"][c]) for c in range(3)]
    oracle.svt_oracle_sao_decide_picture.argtypes = [C.c_void_p] * 4 + [C.c_uint32] * 2 + [C.c_void_p] * 3
    oracle.svt_oracle_sao_decide_picture.restype = None
    oracle.svt_oracle_sao_decide_picture(pic["P"].ctypes.data, st[0].ctypes.data, st[1].ctypes.data, st[2].ctypes.data, pic["cols"],
                                         pic["rows"], pic["enable"].ctypes.data, params.ctypes.data, costs.ctypes.data)
    return params, costs


def stats_of(r):
    st = np.zeros(3, STATS)
    st["boDiff"], st["boCount"], st["eoDiff"], st["eoCount"] = r["bo_diff"], r["bo_count"], r["eo_diff"], r["eo_count"]
    return st


def params_of(r):
    d = np.zeros(1, DEC)
    for k in ("lambda", "chroma_lambda", "type_bits", "merge_bits", "offset_bits", "mm_sao", "temporal_layer"):
        d[k] = r[k]
    d["is_10bit"] = r["is16"]
    return d


def lcu_of(p):
    o = np.zeros(1, LCU)
    for k in ("merge_left", "merge_up", "type", "offset", "band"):
        o[k] = p[k]
    return o


def same_decision(got, want):
    """the reference leaves a switched-off component's offsets and band position as the previous picture had them"""
    if got["merge_left"] != want["merge_left"] or got["merge_up"] != want["merge_up"] or (got["type"] != want["type"]).any():
        return False
    for comp in range(3):
        t = int(want["type"][0 if comp == 0 else 1])
        if t and (got["offset"][comp] != want["offset"][comp]).any():
            return False
        if t == 5 and got["band"][comp] != want["band"][comp]:
            return False
    return True


def oracle_decide(oracle, r):
    st, d, out = stats_of(r), params_of(r), np.zeros(1, LCU)
    left, up = lcu_of(r["left"]), lcu_of(r["up"])
    ptrs = (C.c_void_p * 3)(*[st.ctypes.data + k * STATS.itemsize for k in range(3)])
    costs = np.zeros(2, np.int64)
    oracle.svt_oracle_sao_decide_lcu.argtypes = [C.c_void_p] * 6
    oracle.svt_oracle_sao_decide_lcu.restype = None
    oracle.svt_oracle_sao_decide_lcu(d.ctypes.data, ptrs, left.ctypes.data if r["has_left"] else None,
                                     up.ctypes.data if r["has_up"] else None, out.ctypes.data, costs.ctypes.data)
    return out[0], costs


def test_layouts():
    assert STATS.itemsize == 312 and LCU.itemsize == 72 and DEC.itemsize == 88


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_records(name):
    oracle, recs = S.load_oracle(), load_saodec_case(name)
    assert len(recs) >= 50
    for i, r in enumerate(recs):
        out, costs = oracle_decide(oracle, r)
        assert same_decision(out, r["out"]), (name, i, out, r["out"])
        assert costs[0] == r["luma_cost"], (name, i, costs, r["luma_cost"])
        # the reduced mode of the third and deeper temporal layers returns without touching the costs
        assert costs[1] == r["chroma_cost"], (name, i, costs, r["chroma_cost"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_picture_matches_reference_pictures(name):
    """whole pictures: only the statistics go in, every LCU's neighbours are what the oracle itself decided before"""
    oracle = S.load_oracle()
    pics = pictures_of(name)
    assert pics
    for k, pic in enumerate(pics):
        params, costs = oracle_decide_picture(oracle, pic)
        for i in pic["idx"]:
            assert same_decision(params[i], pic["want"][i]), (name, k, int(i), params[i], pic["want"][i])
        assert (costs[pic["idx"]] == pic["costs"][pic["idx"]]).all()
        off = pic["enable"] == 0
        assert (params["type"][off] == 0).all() and (params["merge_left"][off] == 0).all()


def test_fixture_coverage():
    """every branch of the decision is exercised by some record"""
    recs = np.concatenate([load_saodec_case(n) for n in CASES])
    o = recs["out"]
    assert {0, 1}.issubset(set(recs["mm_sao"].tolist())) and {0, 1}.issubset(set(recs["is16"].tolist()))
    assert set(range(6)).issubset(set(o["type"][:, 0].tolist()))                 # luma: off, four edge classes, band
    assert {0, 1, 2, 3, 4}.issubset(set(o["type"][:, 1].tolist()))               # chroma: off, four edge classes
    assert o["merge_left"].sum() > 20 and o["merge_up"].sum() > 20
    assert ((recs["has_left"] == 0) & (recs["has_up"] == 0)).any()
    ten = recs[recs["is16"] == 1]["out"]
    assert (ten["type"][:, 0] != 0).any() and (ten["type"][:, 1] != 0).any()
