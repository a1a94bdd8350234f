"""-m gpu: encode-pass inter prediction of prediction units (svt_amd_inter_pu_batch, and svt_amd_inter_pu_batch16bit for 10-bit
content) through the C-ABI against (1) records of
real EncodePassInterPrediction calls of the reference with the padded reference pictures they read
(tests/golden/inter_*.npz) and (2) the oracle (pinned to the same records in tests/test_oracle_inter_golden.py) on random
units: every fractional position, vectors far outside the picture (clamp), uni / bi, all sizes."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_inter_golden import CASES, CASES16, JOB, RefPicture, job_of, load_inter_case, ref_struct, want_of

pytestmark = pytest.mark.gpu
vp, u32 = C.c_void_p, C.c_uint32


def run_jobs(product, gpu_ctx, jobs, r0, r1, keep, hbd=False):
    """jobs: JOB array (offsets set here); r0 / r1: RefPicture with DEVICE pointers or None -> list of (y, cb, cr)"""
    import torch
    n = len(jobs)
    jobs = jobs.copy()
    jobs["dst_off_y"] = np.arange(n) * 4096
    jobs["dst_off_c"] = np.arange(n) * 1024
    tdt = torch.int16 if hbd else torch.uint8
    d_y = torch.zeros(n * 4096, dtype=tdt, device="cuda")
    d_cb, d_cr = torch.zeros(n * 1024, dtype=tdt, device="cuda"), torch.zeros(n * 1024, dtype=tdt, device="cuda")
    fn = product.svt_amd_inter_pu_batch16bit if hbd else product.svt_amd_inter_pu_batch
    fn.argtypes = [vp, vp, u32, vp, vp, vp, u32, vp, vp, u32]
    torch.cuda.synchronize()
    out = [None] * n
    # destination strides equal the unit width: launch per width class
    for w in sorted(set(int(v) for v in jobs["pu_w"])):
        idx = np.nonzero(jobs["pu_w"] == w)[0]
        sub = np.ascontiguousarray(jobs[idx])
        rc = fn(gpu_ctx, sub.ctypes.data, len(idx), C.addressof(r0) if r0 else None, C.addressof(r1) if r1 else None, d_y.data_ptr(), w,
                d_cb.data_ptr(), d_cr.data_ptr(), w // 2)
        assert rc == 0, product.svt_amd_last_error()
    hy, hcb, hcr = d_y.cpu().numpy(), d_cb.cpu().numpy(), d_cr.cpu().numpy()
    if hbd:
        hy, hcb, hcr = (a.view(np.uint16) for a in (hy, hcb, hcr))
    for k in range(n):
        w, h = int(jobs["pu_w"][k]), int(jobs["pu_h"][k])
        out[k] = (hy[k * 4096:k * 4096 + w * h].reshape(h, w), hcb[k * 1024:k * 1024 + w * h // 4].reshape(h // 2, w // 2),
                  hcr[k * 1024:k * 1024 + w * h // 4].reshape(h // 2, w // 2))
    return out


def device_refs(g):
    import torch
    keep, refs = [], {}
    for p in g["pic_ids"]:
        t = [torch.from_numpy(np.ascontiguousarray(g["pic%d_%s" % (int(p), c)]).view(np.int16 if g["pred_y"].dtype == np.uint16 else np.uint8)).cuda()
             for c in ("y", "cb", "cr")]
        keep.append(t)
        refs[int(p)] = ref_struct(g, int(p), [x.data_ptr() for x in t])
    return refs, keep


@pytest.mark.parametrize("name,hbd", [(n, False) for n in CASES] + [(n, True) for n in CASES16])
def test_inter_pu_matches_reference_golden(product, gpu_ctx, name, hbd):
    g = load_inter_case(name, hbd)
    refs, keep = device_refs(g)
    n = len(g["pu_w"])
    pairs = sorted(set((int(a), int(b)) for a, b in g["ref_id"]))
    for pa in pairs:     # one batch per pair of reference pictures
        idx = [i for i in range(n) if (int(g["ref_id"][i][0]), int(g["ref_id"][i][1])) == pa]
        jobs = np.concatenate([job_of(g, i) for i in idx])
        got = run_jobs(product, gpu_ctx, jobs, refs.get(pa[0]), refs.get(pa[1]), keep, hbd)
        for k, i in enumerate(idx):
            want = want_of(g, i)
            for p in range(3):
                assert np.array_equal(got[k][p], want[p]), (name, i, p, g["mv"][i].tolist(), int(g["pred_dir"][i]))


@pytest.mark.parametrize("hbd", [False, True])
def test_inter_pu_matches_oracle_random(product, gpu_ctx, oracle, hbd):
    import torch
    ofn = oracle.svt_oracle_inter_pu16bit if hbd else oracle.svt_oracle_inter_pu
    ofn.argtypes = [vp, vp, vp, vp, u32, vp, vp, u32]
    ofn.restype = None
    rng = np.random.default_rng(4 + hbd)
    dt, maxv, bps = (np.uint16, 1024, 2) if hbd else (np.uint8, 256, 1)
    W, H, OX, OY = 192, 128, 68, 68                      # the reference's padding of reference pictures (64 + 4)
    sY, sC = W + 2 * OX, (W + 2 * OX) // 2
    host, dev, keep = [], [], []
    G = 96   # guard rows around the padded picture: the reference's clamp (:802-812) lets the filter taps of a far vector reach
             # a few rows / columns beyond the padding; both sides must then read the same bytes
    for _ in range(2):
        pl = [rng.integers(0, maxv, (H + 2 * OY + 2 * G, sY)).astype(dt), rng.integers(0, maxv, ((H + 2 * OY) // 2 + 2 * G, sC)).astype(dt),
              rng.integers(0, maxv, ((H + 2 * OY) // 2 + 2 * G, sC)).astype(dt)]
        t = [torch.from_numpy(a.view(np.int16) if hbd else a).cuda() for a in pl]
        keep.append((pl, t))
        skip = [G * sY * bps, G * sC * bps, G * sC * bps]
        for store, ptrs in ((host, [a.ctypes.data + k for a, k in zip(pl, skip)]), (dev, [x.data_ptr() + k for x, k in zip(t, skip)])):
            r = RefPicture()
            r.d_y, r.d_cb, r.d_cr = ptrs
            r.strideY, r.strideC, r.originX, r.originY, r.width, r.height = sY, sC, OX, OY, W, H
            store.append(r)
    n = 600
    jobs = np.zeros(n, JOB)
    sizes = rng.choice([8, 16, 32, 64], n)
    jobs["pu_w"], jobs["pu_h"] = sizes, np.where(rng.random(n) < 0.8, sizes, np.maximum(8, sizes // 2))
    jobs["pu_x"] = rng.integers(0, (W - jobs["pu_w"]) // 8 + 1) * 8
    jobs["pu_y"] = rng.integers(0, (H - jobs["pu_h"]) // 8 + 1) * 8
    jobs["pred_dir"] = rng.integers(0, 3, n)
    far = rng.random(n) < 0.25
    jobs["mv"] = np.where(far[:, None, None], rng.integers(-1200, 1200, (n, 2, 2)), rng.integers(-70, 70, (n, 2, 2)))
    got = run_jobs(product, gpu_ctx, jobs, dev[0], dev[1], keep, hbd)
    fracs = set()
    for k in range(n):
        w, h = int(jobs["pu_w"][k]), int(jobs["pu_h"][k])
        want = [np.zeros((h, w), dt), np.zeros((h // 2, w // 2), dt), np.zeros((h // 2, w // 2), dt)]
        ofn(jobs[k:k + 1].ctypes.data, C.addressof(host[0]), C.addressof(host[1]), want[0].ctypes.data, w,
                                   want[1].ctypes.data, want[2].ctypes.data, w // 2)
        for p in range(3):
            assert np.array_equal(got[k][p], want[p]), (k, p, jobs[k])
        fracs.add((int(jobs["mv"][k][0][0]) & 3, int(jobs["mv"][k][0][1]) & 3))
    assert len(fracs) == 16
