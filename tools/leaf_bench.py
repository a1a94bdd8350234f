#!/usr/bin/env python3
"""Throughput of the batched EncDec kernels at picture-sized batches (needs the GPU).
For every kernel: algorithmic bytes (each operand read once, each result written once) / HIP-event time
-> GB/s and the fraction of the 8 TB/s HBM peak.  usage: python tools/leaf_bench.py [iters] [width height]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svtlib as S  # noqa: E402

PEAK = 8000.0
vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    lib = S.load_product()
    ctx = C.c_void_p()
    assert lib.svt_amd_context_create(0, 1920, 1088, 2, C.byref(ctx)) == 0  # the leaf kernels only need its stream
    dev = torch.device("cuda", 0)
    W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
    npx = W * H
    nlcu = ((W + 63) // 64) * ((H + 63) // 64)
    rows = []

    def timed(name, nbytes, fn):
        fn()
        lib.svt_amd_synchronize(ctx)
        torch.cuda.synchronize()
        lib.svt_amd_timer_begin(ctx)
        for _ in range(iters):
            rc = fn()
            assert rc == 0, lib.svt_amd_last_error()
        ms = C.c_float()
        lib.svt_amd_timer_end(ctx, C.byref(ms))
        us = ms.value * 1e3 / iters
        gbs = nbytes / (us * 1e-6) / 1e9
        rows.append({"kernel": name, "us": round(us, 2), "MB": round(nbytes / 1e6, 2), "GB/s": round(gbs, 1),
                     "frac_of_hbm_peak": round(gbs / PEAK, 4)})

    g = torch.Generator(device=dev).manual_seed(1)
    res = torch.randint(-255, 256, (npx,), dtype=torch.int16, device=dev, generator=g)
    coef, rec, q = torch.empty_like(res), torch.empty_like(res), torch.empty_like(res)
    lib.svt_amd_fwd_transform_batch.argtypes = [vp, C.c_int, C.c_int, u32, vp, vp, u32]
    lib.svt_amd_inv_transform_batch.argtypes = [vp, C.c_int, C.c_int, u32, vp, vp, u32]
    lib.svt_amd_quantize_batch.argtypes = [vp, C.c_int, u32, u32, i32, i32, i32, i32, vp, vp, vp, vp, u32]
    lib.svt_amd_full_distortion_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, u32]
    lib.svt_amd_satd_batch.argtypes = [vp, C.c_int, vp, vp, u32]
    # the same 32x32 transforms on the matrix cores (v_mfma_i32_32x32x32_i8, txfm_mfma.hip): north_star's "integer MFMA ... DCT"
    lib.svt_amd_fwd_transform_mfma_batch.argtypes = [vp, C.c_int, C.c_int, u32, vp, vp, u32]
    lib.svt_amd_inv_transform_mfma_batch.argtypes = [vp, C.c_int, u32, vp, vp, u32]
    nb32 = npx // 1024
    timed("fwd_transform 32x32 Estimate, integer MFMA (%d TUs)" % nb32, 4 * nb32 * 1024,
          lambda: lib.svt_amd_fwd_transform_mfma_batch(ctx, 1, 32, 0, res.data_ptr(), coef.data_ptr(), nb32))
    timed("fwd_transform 32x32 full precision, integer MFMA (%d TUs)" % nb32, 4 * nb32 * 1024,
          lambda: lib.svt_amd_fwd_transform_mfma_batch(ctx, 0, 32, 0, res.data_ptr(), coef.data_ptr(), nb32))
    timed("inv_transform 32x32, integer MFMA (%d TUs)" % nb32, 4 * nb32 * 1024,
          lambda: lib.svt_amd_inv_transform_mfma_batch(ctx, 32, 0, coef.data_ptr(), rec.data_ptr(), nb32))
    for size in (32, 16, 8, 4):
        nb = npx // (size * size)
        kind = 1 if size >= 16 else 0
        timed("fwd_transform %dx%d%s (%d TUs)" % (size, size, " Estimate" if kind else "", nb), 4 * nb * size * size,
              lambda: lib.svt_amd_fwd_transform_batch(ctx, kind, size, 0, res.data_ptr(), coef.data_ptr(), nb))
        timed("inv_transform %dx%d (%d TUs)" % (size, size, nb), 4 * nb * size * size,
              lambda: lib.svt_amd_inv_transform_batch(ctx, 0, size, 0, coef.data_ptr(), rec.data_ptr(), nb))
    nz = torch.zeros(npx // 16, dtype=torch.int32, device=dev)
    for size in (32, 8):
        nb = npx // (size * size)
        timed("quantize+inverse %dx%d (%d TUs)" % (size, size, nb), 6 * nb * size * size + 4 * nb,
              lambda: lib.svt_amd_quantize_batch(ctx, size, 26214, 171 << 11, 20, 40 << 2, 1 << 3, 4, coef.data_ptr(),
                                                 q.data_ptr(), rec.data_ptr(), nz.data_ptr(), nb))
    dist = torch.zeros(2 * (npx // 16), dtype=torch.int64, device=dev)
    for size in (32, 8):
        nb = npx // (size * size)
        timed("full_distortion %dx%d (%d TUs)" % (size, size, nb), 4 * nb * size * size + 16 * nb,
              lambda: lib.svt_amd_full_distortion_batch(ctx, size, 0, coef.data_ptr(), rec.data_ptr(), dist.data_ptr(), nb))
    nb = npx // 64
    timed("satd 8x8 (%d blocks)" % nb, 2 * nb * 64 + 8 * nb,
          lambda: lib.svt_amd_satd_batch(ctx, 8, res.data_ptr(), dist.data_ptr(), nb))

    # streaming picture-level kernels
    lib.svt_amd_pack_plane.argtypes = [vp, vp, u32, vp, u32, C.c_int, vp, u32, u32, u32]
    lib.svt_amd_unpack_plane.argtypes = [vp, vp, u32, vp, u32, vp, u32, u32, u32]
    lib.svt_amd_sao_gather_picture.argtypes = [vp, C.c_int, vp, u32, vp, u32, u32, u32, u32, C.c_int, vp]
    in8 = torch.randint(0, 256, (H, W), dtype=torch.uint8, device=dev, generator=g)
    inn = torch.randint(0, 256, (H, W // 4), dtype=torch.uint8, device=dev, generator=g)
    p16 = torch.zeros((H, W), dtype=torch.int16, device=dev)
    o8, on = torch.zeros_like(in8), torch.zeros_like(in8)
    timed("pack 8+2 bit (compressed) -> 16 bit, plane", npx * (1 + 0.25 + 2),
          lambda: lib.svt_amd_pack_plane(ctx, in8.data_ptr(), W, inn.data_ptr(), W // 4, 1, p16.data_ptr(), W, W, H))
    timed("unpack 16 bit -> 8 + 2 bit, plane", npx * (2 + 1 + 1),
          lambda: lib.svt_amd_unpack_plane(ctx, p16.data_ptr(), W, o8.data_ptr(), W, on.data_ptr(), W, W, H))
    rec8 = (in8.to(torch.int16) + torch.randint(-6, 7, (H, W), dtype=torch.int16, device=dev, generator=g)).clamp(0, 255).to(torch.uint8)
    stats = torch.zeros(nlcu * 312, dtype=torch.uint8, device=dev)
    timed("SAO statistics, %d LCUs (BO + 4 EO)" % nlcu, 2 * npx + nlcu * 312,
          lambda: lib.svt_amd_sao_gather_picture(ctx, 1, in8.data_ptr(), W, rec8.data_ptr(), W, W, H, 64, 0, stats.data_ptr()))

    # deblocking: all vertical 8x8-grid luma edges of a plane
    lib.svt_amd_dlf_luma_edges_batch.argtypes = [vp, vp, u32, C.c_int, vp, u32]
    ys, xs = np.meshgrid(np.arange(0, H - 3, 4), np.arange(8, W, 8), indexing="ij")
    e = np.zeros(ys.size, dtype=np.dtype([("offset", "<i4"), ("tc", "<i2"), ("beta", "<i2"), ("v", "u1"), ("pad", "u1", 3)]))
    e["offset"], e["tc"], e["beta"], e["v"] = (ys * W + xs).ravel(), 6, 38, 1
    d_e = torch.from_numpy(e.view(np.uint8)).to(dev)
    plane = in8.clone()
    timed("deblock luma, %d vertical 4-sample edges" % len(e), len(e) * (2 * 32 + 12),
          lambda: lib.svt_amd_dlf_luma_edges_batch(ctx, plane.data_ptr(), W, 1, d_e.data_ptr(), len(e)))

    # whole-picture deblocking (both passes, luma + chroma) and SAO application, 4:2:0
    lib.svt_amd_dlf_picture.argtypes = [vp, C.c_int, vp, u32, vp, vp, u32, u32, u32, vp, vp, vp, u32, i32, i32, i32, i32]
    lib.svt_amd_sao_apply_picture.argtypes = [vp, C.c_int, vp, vp, u32, u32, u32, u32, vp, C.c_int, C.c_int]
    for bps in (1, 2):
        tdt = torch.uint8 if bps == 1 else torch.int16
        hi = 256 if bps == 1 else 1024
        pl = [torch.randint(0, hi, (H >> s, W >> s), dtype=tdt, device=dev, generator=g) for s in (0, 1, 1)]
        out = [torch.zeros_like(t) for t in pl]
        bs = [torch.randint(0, 3, (nlcu, 256), dtype=torch.uint8, device=dev, generator=g) for _ in range(2)]
        qpa = torch.randint(20, 45, ((H // 8) * (W // 8),), dtype=torch.uint8, device=dev, generator=g)
        timed("deblock picture %d-bit (V + H pass, luma + chroma, random strengths)" % (8 if bps == 1 else 10),
              2 * 2 * 1.5 * npx * bps + 2 * nlcu * 512,
              lambda: lib.svt_amd_dlf_picture(ctx, bps, pl[0].data_ptr(), W, pl[1].data_ptr(), pl[2].data_ptr(), W // 2, W, H,
                                              bs[0].data_ptr(), bs[1].data_ptr(), qpa.data_ptr(), W // 8, 0, 0, 0, 0))
        ldt = np.dtype([("ml", "u1"), ("mu", "u1"), ("ef", "u1"), ("pad", "u1"), ("type", "<u4", 2), ("offset", "<i4", (3, 4)),
                        ("band", "<u4", 3)])
        lc = np.zeros(nlcu, ldt)
        r2 = np.random.default_rng(3)
        lc["type"], lc["offset"], lc["band"] = r2.integers(1, 6, (nlcu, 2)), r2.integers(-7, 8, (nlcu, 3, 4)), r2.integers(0, 29, (nlcu, 3))
        d_lc = torch.from_numpy(lc.view(np.uint8)).to(dev)
        ps, pd = (vp * 3)(*[t.data_ptr() for t in pl]), (vp * 3)(*[t.data_ptr() for t in out])
        timed("SAO apply picture %d-bit (all LCUs on, random types)" % (8 if bps == 1 else 10), 2 * 1.5 * npx * bps + nlcu * 76,
              lambda: lib.svt_amd_sao_apply_picture(ctx, bps, ps, pd, W, W // 2, W, H, d_lc.data_ptr(), 1, 1))

    # SAO parameter decision of the picture from per-LCU statistics (own decision per LCU + merge wavefront)
    from test_gpu_saodec import random_picture
    lcols, lrows = (W + 63) // 64, (H + 63) // 64
    sp = random_picture(np.random.default_rng(5), lcols, lrows, 0, 1, 0, 0, 1)
    d_st = [torch.from_numpy(np.ascontiguousarray(sp["stats"][c]).view(np.uint8)).to(dev) for c in range(3)]
    d_sp = torch.from_numpy(sp["params"].view(np.uint8).copy()).to(dev)
    d_sc = torch.zeros((lcols * lrows, 2), dtype=torch.int64, device=dev)
    lib.svt_amd_sao_decide_picture.argtypes = [vp, vp, vp, vp, vp, u32, u32, vp, vp, vp]
    timed("SAO decision picture (%d LCUs: own parameters + merge wavefront)" % (lcols * lrows), lcols * lrows * (3 * 312 + 72 + 16),
          lambda: lib.svt_amd_sao_decide_picture(ctx, sp["P"].ctypes.data, d_st[0].data_ptr(), d_st[1].data_ptr(), d_st[2].data_ptr(), lcols,
                                                 lrows, None, d_sp.data_ptr(), d_sc.data_ptr()))

    # HEVC motion compensation: a 1080p luma plane as 16x16 PUs with random quarter-pel vectors
    lib.svt_amd_mcp_batch_sized.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, u32, vp, u32, vp, u32, u32]
    PADX = 80
    st = W + 2 * PADX
    refp = torch.randint(0, 256, (H + 2 * PADX, st), dtype=torch.uint8, device=dev, generator=g)
    nx, ny = W // 16, H // 16
    rng = np.random.default_rng(0)
    blocks = np.zeros(nx * ny, np.dtype([("ref_off", "<i4"), ("dst_off", "<i4"), ("w", "<u2"), ("h", "<u2"), ("fx", "u1"),
                                         ("fy", "u1"), ("pad", "u1", 2)]))
    mvx, mvy = rng.integers(-64 * 4, 64 * 4, nx * ny), rng.integers(-64 * 4, 64 * 4, nx * ny)
    bx, by = np.tile(np.arange(nx) * 16, ny), np.repeat(np.arange(ny) * 16, nx)
    blocks["ref_off"] = (PADX + by + (mvy >> 2)) * st + PADX + bx + (mvx >> 2)
    blocks["dst_off"], blocks["w"], blocks["h"] = by * W + bx, 16, 16
    blocks["fx"], blocks["fy"] = mvx & 3, mvy & 3
    d_b = torch.from_numpy(blocks.view(np.uint8)).to(dev)
    pred = torch.zeros((H, W), dtype=torch.uint8, device=dev)
    timed("MCP luma uni-pred, %d 16x16 PUs, random 1/4-pel MVs" % len(blocks), 2 * npx,
          lambda: lib.svt_amd_mcp_batch_sized(ctx, 1, 0, 0, refp.data_ptr(), st, pred.data_ptr(), W, d_b.data_ptr(), len(blocks), 16))

    # coefficient rate estimation: every 8x8 / 32x32 TU of a plane, ~10 % non-zero coefficients
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_rate import TU_INFO, synthetic_cost
    lib.svt_amd_coeff_bits_batch.argtypes = [vp, vp, u32, vp, vp, vp, u32]
    cost = synthetic_cost(1)
    for size in (8, 32):
        nb = npx // (size * size)
        q16 = (torch.randint(-3, 4, (npx,), dtype=torch.int16, device=dev, generator=g) *
               (torch.rand(npx, device=dev, generator=g) < 0.1)).to(torch.int16)
        nnz = (q16.view(nb, size * size) != 0).sum(dim=1).cpu().numpy()
        info = np.zeros(nb, TU_INFO)
        info["num_nonzero"], info["type"] = nnz, 1
        d_i = torch.from_numpy(info.view(np.uint8)).to(dev)
        bits = torch.zeros(nb, dtype=torch.int64, device=dev)
        timed("coeff rate estimation %dx%d (%d TUs)" % (size, size, nb), 2 * nb * size * size + 16 * nb,
              lambda: lib.svt_amd_coeff_bits_batch(ctx, cost.ctypes.data, size, q16.data_ptr(), d_i.data_ptr(), bits.data_ptr(), nb))

    # fused mode-decision full loops: one candidate per 32x32 / 16x16 CU of the picture
    from test_oracle_fullloop_golden import FullLoopIn, FullLoopOut
    from test_oracle_chromaloop_golden import ChromaLoopIn, ChromaLoopOut
    lib.svt_amd_full_loop_luma_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32]
    lib.svt_amd_full_loop_chroma_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32]
    for size in (32, 16):
        nc = npx // (size * size)
        fin = np.zeros(nc, np.dtype(FullLoopIn))
        fin["size"], fin["qp"], fin["slice_type"], fin["cand_type"], fin["full_lambda"] = size, 32, 1, 1, 200000
        fin["cbf_bits"] = (20000, 30000, 40000, 20000)
        d_fin = torch.from_numpy(fin.view(np.uint8)).to(dev)
        resid = torch.randint(-20, 21, (nc, 4096), dtype=torch.int16, device=dev, generator=g)
        qo, ro = torch.zeros_like(resid), torch.zeros_like(resid)
        fout = torch.zeros(nc * C.sizeof(FullLoopOut), dtype=torch.uint8, device=dev)
        timed("full loop luma %dx%d (%d candidates: DCT+quant+dist+rate+cost)" % (size, size, nc), nc * (6 * size * size + 72 + 64),
              lambda: lib.svt_amd_full_loop_luma_batch(ctx, cost.ctypes.data, d_fin.data_ptr(), resid.data_ptr(), qo.data_ptr(),
                                                       ro.data_ptr(), fout.data_ptr(), nc))
        # the PM-core variant (encMode 1..4): pm_core = 2 in the upper half of the pf_mode word
        fin["pf_mode"] = 2 << 16
        d_fpm = torch.from_numpy(fin.view(np.uint8)).to(dev)
        lib.svt_amd_full_loop_luma_pmcore_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32]
        timed("full loop luma %dx%d, PM-core quantiser (%d candidates)" % (size, size, nc), nc * (6 * size * size + 72 + 64),
              lambda: lib.svt_amd_full_loop_luma_pmcore_batch(ctx, cost.ctypes.data, d_fpm.data_ptr(), resid.data_ptr(), qo.data_ptr(),
                                                              ro.data_ptr(), fout.data_ptr(), nc))
        cin = np.zeros(nc, np.dtype(ChromaLoopIn))
        cin["size"], cin["cb_qp"], cin["cr_qp"], cin["slice_type"], cin["cand_type"] = size, 31, 31, 1, 1
        d_cin = torch.from_numpy(cin.view(np.uint8)).to(dev)
        cres = torch.randint(-20, 21, (nc, 2048), dtype=torch.int16, device=dev, generator=g)
        cq, cr_ = torch.zeros_like(cres), torch.zeros_like(cres)
        cout = torch.zeros(nc * C.sizeof(ChromaLoopOut), dtype=torch.uint8, device=dev)
        timed("full loop chroma of %dx%d CUs (%d candidates, Cb + Cr)" % (size, size, nc), nc * (6 * size * size // 2 + 32 + 96),
              lambda: lib.svt_amd_full_loop_chroma_batch(ctx, cost.ctypes.data, d_cin.data_ptr(), cres.data_ptr(), cq.data_ptr(),
                                                         cr_.data_ptr(), cout.data_ptr(), nc))

    # the whole transform unit of the final encode pass in one kernel (residual -> T -> Q -> iQ -> iT -> recon), luma plane
    lib.svt_amd_encode_tu_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, u32, vp, u32, vp, vp, u32]
    eudt = np.dtype([("src_off", "<i4"), ("rec_off", "<i4"), ("qp", "u1"), ("slice_type", "u1"), ("pad", "u1", 2), ("dz", "<u4")])
    srcp = torch.randint(0, 256, (H, W), dtype=torch.uint8, device=dev, generator=g)
    for size in (32, 16, 8):
        xs, ys = np.meshgrid(np.arange(0, W - size + 1, size), np.arange(0, H - size + 1, size))
        eu = np.zeros(xs.size, eudt)
        eu["src_off"] = eu["rec_off"] = (ys * W + xs).ravel()
        eu["qp"], eu["slice_type"] = 32, 1
        d_eu = torch.from_numpy(eu.view(np.uint8)).to(dev)
        recp = (srcp.to(torch.int16) + torch.randint(-12, 13, (H, W), dtype=torch.int16, device=dev, generator=g)).clamp(0, 255).to(torch.uint8)
        qo2 = torch.zeros(len(eu) * size * size, dtype=torch.int16, device=dev)
        nz2 = torch.zeros(len(eu), dtype=torch.int32, device=dev)
        timed("encode TU %dx%d fused (%d units: residual+DCT+quant+iquant+iDCT+recon)" % (size, size, len(eu)),
              len(eu) * (size * size * 5 + 16 + 4),
              lambda: lib.svt_amd_encode_tu_batch(ctx, 1, size, d_eu.data_ptr(), srcp.data_ptr(), W, recp.data_ptr(), W, qo2.data_ptr(),
                                                  nz2.data_ptr(), len(eu)))

    print(json.dumps({"iters": iters, "width": W, "height": H, "peak_GBs": PEAK, "kernels": rows}, indent=1))
    lib.svt_amd_context_destroy(ctx)


if __name__ == "__main__":
    main()
