"""-m gpu: the distortion stage of the fast loop (svt_amd_fast_loop_distortion_batch) through the C-ABI against (1) records of
real second-loop iterations of ProductPerformFastLoop (tests/golden/fastloop_*.npz) and (2) the oracle (pinned to the same
records in tests/test_oracle_fastloop_golden.py) on a 1080p picture: every CU size, unaligned strides, chroma on / off,
most-probable-mode candidates."""
import ctypes as C

import numpy as np
import pytest

import svtlib as S
from test_oracle_fastloop_golden import CAND, CASES, DIST, blocks_of, cand_of, load_fastloop_case, reference_rule

pytestmark = pytest.mark.gpu
vp, u32 = C.c_void_p, C.c_uint32
ARGS = [vp, vp, u32, vp, vp, u32, vp, u32, vp, vp, u32, vp, u32, vp]


def run(product, gpu_ctx, planes_src, planes_pred, strides, cands):
    import torch
    d = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in planes_src + planes_pred]
    d_c = torch.from_numpy(cands.view(np.uint8).copy()).cuda()
    d_o = torch.zeros(len(cands) * 8, dtype=torch.uint8, device="cuda")
    product.svt_amd_fast_loop_distortion_batch.argtypes = ARGS
    torch.cuda.synchronize()
    rc = product.svt_amd_fast_loop_distortion_batch(gpu_ctx, d[0].data_ptr(), strides[0], d[1].data_ptr(), d[2].data_ptr(), strides[1],
                                                    d[3].data_ptr(), strides[2], d[4].data_ptr(), d[5].data_ptr(), strides[3],
                                                    d_c.data_ptr(), len(cands), d_o.data_ptr())
    assert rc == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    return d_o.cpu().numpy().view(DIST)


@pytest.mark.parametrize("name", CASES)
def test_fast_loop_distortion_matches_reference_golden(product, gpu_ctx, name):
    g = load_fastloop_case(name)
    n = len(g["size"])
    # all records side by side in six long planes: record i owns a 64-sample wide column strip
    sy, py = np.zeros((64, 64 * n), np.uint8), np.zeros((64, 64 * n), np.uint8)
    sc, pc = [np.zeros((32, 32 * n), np.uint8) for _ in range(2)], [np.zeros((32, 32 * n), np.uint8) for _ in range(2)]
    cands = np.zeros(n, CAND)
    for i in range(n):
        src, pred = blocks_of(g, i)
        z = int(g["size"][i])
        sy[:z, 64 * i:64 * i + z], py[:z, 64 * i:64 * i + z] = src[0], pred[0]
        for p in range(2):
            sc[p][:z // 2, 32 * i:32 * i + z // 2], pc[p][:z // 2, 32 * i:32 * i + z // 2] = src[1 + p], pred[1 + p]
        cands[i] = cand_of(g, i)[0]
        cands[i]["src_off_y"] = cands[i]["pred_off_y"] = 64 * i
        cands[i]["src_off_c"] = cands[i]["pred_off_c"] = 32 * i
    out = run(product, gpu_ctx, [sy, sc[0], sc[1]], [py, pc[0], pc[1]], [64 * n, 32 * n, 64 * n, 32 * n], cands)
    for i in range(n):
        want = (int(g["luma_distortion"][i]), int(g["chroma_distortion"][i]))
        got = (int(out[i]["luma"]), int(out[i]["chroma"]))
        assert got == want or reference_rule(g, i, out[i]) == want, (name, i, got, want)


@pytest.mark.parametrize("pad", [0, 3])
def test_fast_loop_distortion_matches_oracle_random(product, gpu_ctx, oracle, pad):
    oracle.svt_oracle_fast_loop_distortion.argtypes = [vp, vp, u32, vp, vp, u32, vp, u32, vp, vp, u32, vp]
    oracle.svt_oracle_fast_loop_distortion.restype = None
    rng = np.random.default_rng(11 + pad)
    W, H = 1920, 1080
    src = [rng.integers(0, 256, (H, W + pad), dtype=np.uint8), rng.integers(0, 256, (H // 2, W // 2 + pad), dtype=np.uint8),
           rng.integers(0, 256, (H // 2, W // 2 + pad), dtype=np.uint8)]
    pred = [np.clip(a.astype(np.int16) + rng.integers(-9, 10, a.shape), 0, 255).astype(np.uint8) for a in src]
    n = 3000
    cands = np.zeros(n, CAND)
    cands["size"] = rng.choice([8, 16, 32, 64], n)
    x = rng.integers(0, (W - 64) // 8, n) * 8 + (rng.integers(0, 2, n) * 2 if pad else 0)
    y = rng.integers(0, (H - 64) // 8, n) * 8
    cands["src_off_y"], cands["src_off_c"] = y * (W + pad) + x, (y // 2) * (W // 2 + pad) + x // 2
    x2, y2 = rng.integers(0, (W - 64) // 2, n) * 2, rng.integers(0, (H - 64) // 2, n) * 2
    cands["pred_off_y"], cands["pred_off_c"] = y2 * (W + pad) + x2, (y2 // 2) * (W // 2 + pad) + x2 // 2
    cands["flags"] = rng.integers(0, 2, n) | (rng.random(n) < 0.05) * 2
    out = run(product, gpu_ctx, src, pred, [W + pad, W // 2 + pad, W + pad, W // 2 + pad], cands)
    for k in range(n):
        d = np.zeros(1, DIST)
        oracle.svt_oracle_fast_loop_distortion(cands[k:k + 1].ctypes.data, src[0].ctypes.data, W + pad, src[1].ctypes.data, src[2].ctypes.data,
                                               W // 2 + pad, pred[0].ctypes.data, W + pad, pred[1].ctypes.data, pred[2].ctypes.data,
                                               W // 2 + pad, d.ctypes.data)
        assert out[k] == d[0], (k, cands[k], out[k], d[0])
    assert (out["luma"] == 0).sum() >= 50 and (out["chroma"] > 0).sum() > 1000


def test_fast_loop_distortion_rejects_bad_arguments(product, gpu_ctx):
    product.svt_amd_fast_loop_distortion_batch.argtypes = ARGS
    assert product.svt_amd_fast_loop_distortion_batch(gpu_ctx, None, 64, None, None, 0, 8, 64, None, None, 0, 8, 1, 8) != 0
    assert product.svt_amd_fast_loop_distortion_batch(gpu_ctx, 8, 64, 8, None, 32, 8, 64, None, None, 0, 8, 1, 8) != 0
    assert product.svt_amd_fast_loop_distortion_batch(gpu_ctx, 8, 64, None, None, 0, 8, 64, None, None, 0, 8, 0, 8) != 0


def test_predict_all_then_measure_all_on_the_device(product, gpu_ctx, oracle):
    """The fast loop's data-parallel part as two launches with nothing crossing PCIe in between: svt_amd_intra_pu_batch predicts every
    candidate (neighbour slices of real mode-decision records) into device planes, svt_amd_fast_loop_distortion_batch measures them
    against a source picture; checked against the oracle's prediction + SADs."""
    import torch
    from test_oracle_intra_golden import JOB
    from test_oracle_intramd_golden import CASES as MD_CASES, load_intramd_case, md_job_of
    g = load_intramd_case(MD_CASES[0])
    idx = [i for i in range(len(g["size"])) if int(g["component_mask"][i]) == 1][:120]
    n = len(idx)
    jobs = np.concatenate([md_job_of(g, i) for i in idx])
    jobs["dst_off_y"] = np.arange(n) * 32            # candidate k owns columns [32k, 32k + size) of a 32-row prediction plane
    jobs["dst_off_c"] = np.arange(n) * 16
    rng = np.random.default_rng(8)
    srcY = rng.integers(0, 256, (32, 32 * n), dtype=np.uint8)
    srcC = [rng.integers(0, 256, (16, 16 * n), dtype=np.uint8) for _ in range(2)]
    d_jobs = torch.from_numpy(jobs.view(np.uint8).copy()).cuda()
    d_py = torch.zeros((32, 32 * n), dtype=torch.uint8, device="cuda")
    d_pc = [torch.zeros((16, 16 * n), dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_src = [torch.from_numpy(a).cuda() for a in [srcY] + srcC]
    cands = np.zeros(n, CAND)
    cands["size"], cands["flags"] = jobs["size"], 1
    cands["src_off_y"] = cands["pred_off_y"] = jobs["dst_off_y"]
    cands["src_off_c"] = cands["pred_off_c"] = jobs["dst_off_c"]
    d_c = torch.from_numpy(cands.view(np.uint8).copy()).cuda()
    d_o = torch.zeros(n * 8, dtype=torch.uint8, device="cuda")
    product.svt_amd_intra_pu_batch.argtypes = [vp, C.c_int, vp, u32, vp, u32, vp, vp, u32]
    product.svt_amd_fast_loop_distortion_batch.argtypes = ARGS
    torch.cuda.synchronize()
    # the kernel takes one row pitch per launch: all candidates share the 32 * n wide planes
    assert product.svt_amd_intra_pu_batch(gpu_ctx, 1, d_jobs.data_ptr(), n, d_py.data_ptr(), 32 * n, d_pc[0].data_ptr(), d_pc[1].data_ptr(),
                                          16 * n) == 0, product.svt_amd_last_error()
    assert product.svt_amd_fast_loop_distortion_batch(gpu_ctx, d_src[0].data_ptr(), 32 * n, d_src[1].data_ptr(), d_src[2].data_ptr(), 16 * n,
                                                      d_py.data_ptr(), 32 * n, d_pc[0].data_ptr(), d_pc[1].data_ptr(), 16 * n, d_c.data_ptr(),
                                                      n, d_o.data_ptr()) == 0, product.svt_amd_last_error()
    product.svt_amd_synchronize(gpu_ctx)
    out = d_o.cpu().numpy().view(DIST)
    oracle.svt_oracle_intra_pu.argtypes = [C.c_int, vp, vp, u32, vp, vp, u32]
    oracle.svt_oracle_intra_pu.restype = None
    for k in range(n):
        z = int(jobs["size"][k])
        p = [np.zeros((z, z), np.uint8), np.zeros((z // 2, z // 2), np.uint8), np.zeros((z // 2, z // 2), np.uint8)]
        oracle.svt_oracle_intra_pu(1, jobs[k:k + 1].ctypes.data, p[0].ctypes.data, z, p[1].ctypes.data, p[2].ctypes.data, z // 2)
        luma = int(np.abs(srcY[:z, 32 * k:32 * k + z].astype(int) - p[0]).sum())
        chroma = sum(int(np.abs(srcC[c][:z // 2, 16 * k:16 * k + z // 2].astype(int) - p[1 + c]).sum()) for c in range(2))
        assert (int(out[k]["luma"]), int(out[k]["chroma"])) == (luma, chroma), (k, z, out[k], luma, chroma)
