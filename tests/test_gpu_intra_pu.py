"""-m gpu: fused encode-pass intra prediction of a prediction unit (svt_amd_intra_pu_batch / svt_amd_intra_pu) through the
C-ABI against (1) records of real GenerateIntraReferenceSamplesEncodePass + EncodePassIntraPrediction call pairs of the
reference (tests/golden/intra_*.npz) and (2) the oracle (pinned to the same records in tests/test_oracle_intra_golden.py) on
random neighbourhoods: every mode and size, random availability patterns, constrained intra, 8 and 10 bit."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_intra_golden import CASES, JOB, job_of, load_intra_case, want_of

pytestmark = pytest.mark.gpu
vp, u32 = C.c_void_p, C.c_uint32


def run_jobs(product, gpu_ctx, bps, jobs):
    """jobs: JOB array with dst offsets laid out by this function -> list of (y, cb, cr) blocks"""
    import torch
    n = len(jobs)
    jobs = jobs.copy()
    jobs["dst_off_y"] = np.arange(n) * 1024
    jobs["dst_off_c"] = np.arange(n) * 256
    tdt = torch.uint8 if bps == 1 else torch.int16
    d_j = torch.from_numpy(jobs.view(np.uint8).copy()).cuda()
    d_y = torch.zeros(n * 1024, dtype=tdt, device="cuda")
    d_cb, d_cr = torch.zeros(n * 256, dtype=tdt, device="cuda"), torch.zeros(n * 256, dtype=tdt, device="cuda")
    product.svt_amd_intra_pu_batch.argtypes = [vp, C.c_int, vp, u32, vp, u32, vp, vp, u32]
    out = []
    torch.cuda.synchronize()
    # strides differ per size: launch per size class with its own stride
    for size in (8, 16, 32):
        idx = np.nonzero(jobs["size"] == size)[0]
        if not len(idx):
            continue
        sub = torch.from_numpy(jobs[idx].view(np.uint8).copy()).cuda()
        rc = product.svt_amd_intra_pu_batch(gpu_ctx, bps, sub.data_ptr(), len(idx), d_y.data_ptr(), size, d_cb.data_ptr(),
                                            d_cr.data_ptr(), size // 2)
        assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    hy, hcb, hcr = (a.cpu().numpy() for a in (d_y, d_cb, d_cr))
    if bps == 2:
        hy, hcb, hcr = (a.view(np.uint16) for a in (hy, hcb, hcr))
    for k in range(n):
        s = int(jobs["size"][k])
        c = s // 2
        out.append((hy[k * 1024:k * 1024 + s * s].reshape(s, s), hcb[k * 256:k * 256 + c * c].reshape(c, c),
                    hcr[k * 256:k * 256 + c * c].reshape(c, c)))
    return out


def test_job_size(product):
    assert JOB.itemsize == 832


@pytest.mark.parametrize("name", CASES)
def test_intra_pu_matches_reference_golden(product, gpu_ctx, name):
    g = load_intra_case(name)
    idx = [i for i in range(len(g["size"])) if int(g["size"][i]) <= 32]
    bps = int(g["bytes_per_sample"][idx[0]])
    jobs = np.concatenate([job_of(g, i) for i in idx])
    got = run_jobs(product, gpu_ctx, bps, jobs)
    for k, i in enumerate(idx):
        want = want_of(g, i)
        for p in range(3):
            assert np.array_equal(got[k][p], want[p]), (name, i, p, int(g["size"][i]), int(g["luma_mode"][i]))


def test_intra_pu_matches_mode_decision_records(product, gpu_ctx):
    """records of the mode decision's IntraPredictionCl: batched form for all of them, and the per-call host form asking for
    the luma block or the chroma pair alone (what the binding of IntraPredictionCl does)"""
    from test_oracle_intramd_golden import CASES as MD_CASES, OL_CASES, load_intramd_case, md_job_of, planes_of
    product.svt_amd_intra_pu.argtypes = [vp, C.c_int, vp, vp, u32, vp, vp, u32]
    assert len(MD_CASES) == 4 and len(OL_CASES) == 3
    for name, ol in [(n, False) for n in MD_CASES] + [(n, True) for n in OL_CASES]:     # closed loop, then the open-loop twin
        g = load_intramd_case(name, ol)
        n = len(g["size"])
        jobs = np.concatenate([md_job_of(g, i) for i in range(n)])
        got = run_jobs(product, gpu_ctx, 1, jobs)
        for i in range(n):
            want = want_of(g, i)
            for p in planes_of(int(g["component_mask"][i])):
                assert np.array_equal(got[i][p], want[p]), (name, i, p, int(g["size"][i]), int(g["luma_mode"][i]))
        for i in range(0, n, 5):
            want, j = want_of(g, i), md_job_of(g, i)
            s_, c_ = int(g["size"][i]), int(g["size"][i]) // 2
            if int(g["component_mask"][i]) == 1:
                y = np.full((s_, 64), 0xAA, np.uint8)
                rc = product.svt_amd_intra_pu(gpu_ctx, 1, j.ctypes.data, y.ctypes.data, 64, None, None, 0)
                assert rc == 0 and np.array_equal(y[:, :s_], want[0]) and (y[:, s_:] == 0xAA).all(), (name, i)
            else:
                cb, cr = np.full((c_, 32), 0xAA, np.uint8), np.full((c_, 32), 0xAA, np.uint8)
                rc = product.svt_amd_intra_pu(gpu_ctx, 1, j.ctypes.data, None, 0, cb.ctypes.data, cr.ctypes.data, 32)
                assert rc == 0 and np.array_equal(cb[:, :c_], want[1]) and np.array_equal(cr[:, :c_], want[2]), (name, i)
                assert (cb[:, c_:] == 0xAA).all()
    j = job_of(load_intramd_case(MD_CASES[0]), 0)
    assert product.svt_amd_intra_pu(gpu_ctx, 1, j.ctypes.data, None, 0, None, None, 0) != 0
    assert product.svt_amd_intra_pu(gpu_ctx, 1, j.ctypes.data, None, 0, j.ctypes.data, None, 32) != 0


def test_intra_pu_matches_intra4x4_records(product, gpu_ctx):
    """intra 4x4 coding units of the encode pass: size-4 jobs (luma partition) and size-8 jobs (the coding unit's chroma pair),
    through the per-call host form the binding uses"""
    from test_oracle_intra4_golden import CASES as I4_CASES, load_intra4_case
    product.svt_amd_intra_pu.argtypes = [vp, C.c_int, vp, vp, u32, vp, vp, u32]
    assert len(I4_CASES) == 3
    for name in I4_CASES:
        g = load_intra4_case(name)
        for i in range(len(g["size"])):
            bps, mask = int(g["bytes_per_sample"][i]), int(g["component_mask"][i])
            dt = np.uint8 if bps == 1 else np.uint16
            want, j = want_of(g, i), job_of(g, i)
            if mask == 1:
                y = np.full((4, 16), 0xAA, dt)
                rc = product.svt_amd_intra_pu(gpu_ctx, bps, j.ctypes.data, y.ctypes.data, 16, None, None, 0)
                assert rc == 0 and np.array_equal(y[:, :4], want[0]) and (y[:, 4:] == 0xAA).all(), (name, i, int(g["luma_mode"][i]))
            else:
                cb, cr = np.full((4, 8), 0xAA, dt), np.full((4, 8), 0xAA, dt)
                rc = product.svt_amd_intra_pu(gpu_ctx, bps, j.ctypes.data, None, 0, cb.ctypes.data, cr.ctypes.data, 8)
                assert rc == 0 and np.array_equal(cb[:, :4], want[1]) and np.array_equal(cr[:, :4], want[2]), (name, i)
    j = job_of(load_intra4_case(I4_CASES[0]), 0)
    j["size"] = 4
    buf = np.zeros((8, 8), np.uint8)
    assert product.svt_amd_intra_pu(gpu_ctx, 1, j.ctypes.data, buf.ctypes.data, 8, buf.ctypes.data, buf.ctypes.data, 8) != 0   # 4x4: luma only


def random_jobs(rng, n, bps):
    maxv = 255 if bps == 1 else 1023
    jobs = np.zeros(n, JOB)
    jobs["size"] = rng.choice([8, 16, 32], n)
    jobs["luma_mode"] = np.arange(n) % 35
    jobs["chroma_mode"] = rng.integers(0, 5, n)
    for f in ("constrained_intra", "strong_smoothing", "bottom_left_ok", "top_right_ok"):
        jobs[f] = rng.integers(0, 2, n)
    for f in ("pic_left", "pic_top", "pic_right"):
        jobs[f] = rng.random(n) < 0.15
    kinds = np.array([1, 2, 2, 2, 0xFF, 0xFE], np.uint8)
    jobs["mode_left"], jobs["mode_top"] = kinds[rng.integers(0, 6, (n, 16))], kinds[rng.integers(0, 6, (n, 16))]
    jobs["mode_tl"] = kinds[rng.integers(0, 5, n)]
    smooth = rng.random(n) < 0.4     # flat neighbourhoods so that the strong filter fires
    base = rng.integers(0, maxv + 1, (n, 1, 1))
    jobs["left"] = np.where(smooth[:, None, None], np.clip(base + rng.integers(-2, 3, (n, 3, 64)), 0, maxv), rng.integers(0, maxv + 1, (n, 3, 64)))
    jobs["top"] = np.where(smooth[:, None, None], np.clip(base + rng.integers(-2, 3, (n, 3, 64)), 0, maxv), rng.integers(0, maxv + 1, (n, 3, 64)))
    jobs["tl"] = np.where(smooth[:, None], np.clip(base[:, 0] + rng.integers(-2, 3, (n, 3)), 0, maxv), rng.integers(0, maxv + 1, (n, 3)))
    return jobs


@pytest.mark.parametrize("bps", [1, 2])
def test_intra_pu_matches_oracle_random(product, gpu_ctx, oracle, bps):
    oracle.svt_oracle_intra_pu.argtypes = [C.c_int, vp, vp, u32, vp, vp, u32]
    oracle.svt_oracle_intra_pu.restype = None
    rng = np.random.default_rng(5 + bps)
    jobs = random_jobs(rng, 700, bps)
    got = run_jobs(product, gpu_ctx, bps, jobs)
    dt = np.uint8 if bps == 1 else np.uint16
    none_avail = 0
    for k in range(len(jobs)):
        s = int(jobs["size"][k])
        want = [np.zeros((s, s), dt), np.zeros((s // 2, s // 2), dt), np.zeros((s // 2, s // 2), dt)]
        oracle.svt_oracle_intra_pu(bps, jobs[k:k + 1].ctypes.data, want[0].ctypes.data, s, want[1].ctypes.data, want[2].ctypes.data, s // 2)
        for p in range(3):
            assert np.array_equal(got[k][p], want[p]), (k, p, s, int(jobs["luma_mode"][k]), np.argwhere(got[k][p] != want[p])[:4].tolist())
        none_avail += int((want[0] == (128 if bps == 1 else 512)).all())
    assert none_avail >= 1   # the all-unavailable corner occurred


def test_intra_pu_host_pointer_form(product, gpu_ctx, oracle):
    oracle.svt_oracle_intra_pu.argtypes = [C.c_int, vp, vp, u32, vp, vp, u32]
    oracle.svt_oracle_intra_pu.restype = None
    product.svt_amd_intra_pu.argtypes = [vp, C.c_int, vp, vp, u32, vp, vp, u32]
    rng = np.random.default_rng(9)
    jobs = random_jobs(rng, 30, 1)
    for k in range(len(jobs)):
        s = int(jobs["size"][k])
        want = [np.zeros((s, 64), np.uint8), np.zeros((s // 2, 32), np.uint8), np.zeros((s // 2, 32), np.uint8)]
        got = [np.full_like(w, 9) for w in want]
        for w in want:
            w[:] = 9
        oracle.svt_oracle_intra_pu(1, jobs[k:k + 1].ctypes.data, want[0].ctypes.data, 64, want[1].ctypes.data, want[2].ctypes.data, 32)
        rc = product.svt_amd_intra_pu(gpu_ctx, 1, jobs[k:k + 1].ctypes.data, got[0].ctypes.data, 64, got[1].ctypes.data,
                                      got[2].ctypes.data, 32)
        assert rc == 0, product.svt_amd_last_error()
        for p in range(3):
            assert np.array_equal(got[p], want[p]), (k, p)
