#!/usr/bin/env python3
"""bench.py - hot-path throughput of the MI355X-native SVT-HEVC block-analysis path.

One "step" = one pass of the hot path over one batch of synthetic 1080p pictures
(BASELINE.json configs[1]: 1920x1080 8-bit, encMode 9, low-delay P): for every picture
of the batch, picture preparation (pad + 1/4 + 1/16 decimation + half-pel planes),
open-loop motion estimation of all 510 LCUs against the previous picture (HME L0/L1,
full-pel 85-PU search, half/quarter-pel refinement, candidate records) and open-loop
intra search (OIS points from the ME distortions, stage-1 modes, candidate injection),
with the controls exactly as the reference encoder derived them for this configuration
(tests/golden/me_p_1920x1080_m9.npz, ois_ip_1920x1080_m9.npz).  Inputs are resident in HBM before the timed
region; results stay in HBM (the PCIe-inclusive rate is discussed in DESIGN.md).

Prints ONE JSON line (rank 0).  `value` = pictures/s of the hot path over all GPUs,
NOT whole-encoder fps: the closed-loop EncDec half is not on the device yet
(DESIGN.md "scope").  N>1: pictures are sharded over ranks (independent, no collective).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svtlib as S  # noqa: E402
from golden_util import load_case  # noqa: E402

W, H = 1920, 1080
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def synth_frames_device(n, seed, device):
    """Moving-texture luma frames generated on the device (uint8 [n, H, W])."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    noise = torch.randint(0, 256, (H + 2 * n + 8, W + 4 * n + 8), dtype=torch.uint8, device=device, generator=g)
    x = torch.arange(W, device=device, dtype=torch.float32)[None, :]
    y = torch.arange(H, device=device, dtype=torch.float32)[:, None]
    frames = torch.empty((n, H, W), dtype=torch.uint8, device=device)
    for t in range(n):
        l = 128 + 50 * (torch.sin((x + 3 * t) / 17.0) + torch.cos((y + 2 * t) / 23.0)) + \
            (noise[t:t + H, 2 * t:2 * t + W] >> 4).float()
        frames[t] = l.clamp(0, 255).to(torch.uint8)
    return frames


def cpu_baseline_port(params, oparams, budget_s=12.0):
    """Oracle (plain C restatement, 1 thread) on a bounded sample of the same workload: whole pictures
    (prep + ME + OIS + the residual / DCT / quantiser / reconstruction stage each) until the time budget is used."""
    oracle = S.load_oracle()
    oracle.svt_oracle_encode_plane.restype = C.c_uint64
    oracle.svt_oracle_encode_plane.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 7
    H16 = H // 16 * 16
    frames = [S.gen_luma("motion", W, H, t, 7) for t in range(4)]
    nl = S.lcu_count(W, H)
    prev = S.OraclePicture(oracle, frames[0])
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        f = frames[(done + 1) % len(frames)]
        cur = S.OraclePicture(oracle, f)                                  # pad + decimate + half-pel planes
        me = S.oracle_me_picture(oracle, params, cur, prev, None, 0, nl)  # ME of all LCUs
        S.oracle_ois_picture(oracle, oparams, f, me)                      # OIS of all LCUs
        src, rec = np.ascontiguousarray(f), np.ascontiguousarray(frames[done % len(frames)]).copy()
        oracle.svt_oracle_encode_plane(src.ctypes.data, rec.ctypes.data, W, W, 0, H16, 16, 32, 1)          # 16x16 units
        oracle.svt_oracle_encode_plane(src.ctypes.data, rec.ctypes.data, W, W, H16, H - H16, 8, 32, 1)     # last 8 rows: 8x8
        prev = cur
        done += 1
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 4), "unit": "fps", "cores": 1, "kind": "port",
            "sample": "%d whole 1080p P pictures (prep + ME + OIS of 510 LCUs + residual/DCT/quantiser/reconstruction of the luma plane "
                      "each; oracle/svt_oracle_me.c, svt_oracle_ois.c, svt_oracle_fullloop.c:svt_oracle_encode_plane; 1 thread; "
                      "%.1f s)" % (done, dt)}


def reference_encoder_fps(frames=24):
    """Whole reference encoder (AVX2 path, all host cores) on the same configuration - context only."""
    if not os.path.exists(S.REF_APP):
        return None
    try:
        with tempfile.TemporaryDirectory() as td:
            yuv = os.path.join(td, "c.yuv")
            S.write_clip(yuv, "motion", W, H, frames, 7)
            out = subprocess.run([S.REF_APP, "-i", yuv, "-w", str(W), "-h", str(H), "-n", str(frames), "-encMode", "9",
                                  "-pred-struct", "0", "-q", "32", "-asm", "1", "-b", os.path.join(td, "o.265")],
                                 capture_output=True, text=True, timeout=300).stdout
        for line in out.splitlines():
            if "Average Speed" in line:
                return {"value": float(line.split()[2]), "unit": "fps (whole encoder, -asm 1)",
                        "cores": os.cpu_count(), "frames": frames}
    except Exception as e:  # context only
        return {"error": str(e)}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="pictures per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent picture sequences per GPU, each on its own context/stream, stepped round-robin "
                         "(2 overlaps the latency-bound kernels of neighbouring batches; per-kernel times then inflate)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    lib = S.load_product()  # fails loudly when the HIP library is absent
    B = a.batch
    g = load_case("p_1920x1080_m9")
    params = S.params_from_record(g["params"][0])
    og = np.load(os.path.join(S.GOLDEN_DIR, "ois_ip_1920x1080_m9.npz"))
    oparams = S.ois_params_from_record(og["params"][1])  # the P picture of the reference run
    assert not oparams.slice_is_intra
    lib.svt_amd_picture_upload_device_batch.restype = C.c_int
    lib.svt_amd_picture_upload_device_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p),
                                                        C.c_uint32, C.c_uint16, C.c_uint16]
    # Ring of B+1 slots: picture n lives in slot n % (B+1) and is searched against picture n-1, so every
    # picture is prepared exactly once and the B pictures of a step are independent of each other
    # (the front half is open loop): per step ONE prep launch, ONE ME batch (2 launches), ONE OIS launch.
    R = B + 1
    # Residual / DCT / quantiser / reconstruction stage (the "+ DCT" of BASELINE configs[1]): every picture's luma plane as 16x16
    # transform units (8x8 for the last 8 rows of 1080), each predicted from the co-located block of the working reconstruction
    # plane of its batch position, through the fused encode-pass kernel (svt_amd_encode_tu_batch: residual -> DCT -> Q -> iQ ->
    # iDCT -> reconstruction in place).  One launch per unit size and step.
    eudt = np.dtype([("src_off", "<i4"), ("rec_off", "<i4"), ("qp", "u1"), ("slice_type", "u1"), ("pad", "u1", 2), ("dz", "<u4")])
    lib.svt_amd_encode_tu_batch.restype = C.c_int
    lib.svt_amd_encode_tu_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.c_uint32]
    H16 = H // 16 * 16
    g16 = np.stack(np.meshgrid(np.arange(0, H16, 16), np.arange(0, W, 16), indexing="ij"), -1).reshape(-1, 2)
    g8 = np.stack(np.meshgrid(np.arange(H16, H, 8), np.arange(0, W, 8), indexing="ij"), -1).reshape(-1, 2)

    def unit_list(grid, cur_slots):
        u = np.zeros(len(cur_slots) * len(grid), eudt)
        for i, cur in enumerate(cur_slots):
            blk = u[i * len(grid):(i + 1) * len(grid)]
            blk["src_off"] = cur * H * W + grid[:, 0] * W + grid[:, 1]
            blk["rec_off"] = i * H * W + grid[:, 0] * W + grid[:, 1]
        u["qp"], u["slice_type"] = 32, 1
        return u

    def make_lane(idx):
        ctx = C.c_void_p()
        rc = lib.svt_amd_context_create(local_rank, W, H + 8, R, C.byref(ctx))
        assert rc == 0, lib.svt_amd_last_error()
        frames = synth_frames_device(R, 1234 + 17 * rank + idx, dev)
        rec = frames[:B].clone()                                     # working reconstruction planes, one per batch position
        quant = torch.zeros((B, H, W), dtype=torch.int16, device=dev)
        nz = torch.zeros(B * (len(g16) + len(g8)), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        phases = []
        for k in range(R):  # step s starts at picture n0 = 1 + s*B; phase = n0 % R
            slots = (C.c_int * B)()
            ptrs = (C.c_void_p * B)()
            jobs, ojobs = (S.MeJob * B)(), (S.OisJob * B)()
            for i in range(B):
                cur, ref = (k + i) % R, (k + i - 1) % R
                slots[i], ptrs[i] = cur, frames[cur].data_ptr()
                jobs[i].params, jobs[i].cur_slot = params, cur
                jobs[i].ref_slot[0] = jobs[i].ref_slot[1] = ref
                ojobs[i].params, ojobs[i].cur_slot = oparams, cur
            curs = [(k + i) % R for i in range(B)]
            u16 = torch.from_numpy(unit_list(g16, curs).view(np.uint8)).to(dev)
            u8 = torch.from_numpy(unit_list(g8, curs).view(np.uint8)).to(dev) if len(g8) else None
            phases.append((slots, ptrs, jobs, ojobs, u16, u8))
        state = {"n0": 1}

        def step():
            slots, ptrs, jobs, ojobs, u16, u8 = phases[state["n0"] % R]
            r = lib.svt_amd_picture_upload_device_batch(ctx, B, slots, ptrs, W, W, H)
            assert r == 0, lib.svt_amd_last_error()
            r = lib.svt_amd_me_batch_launch(ctx, jobs, B)
            assert r == 0, lib.svt_amd_last_error()
            r = lib.svt_amd_ois_batch_launch(ctx, ojobs, B)  # reads the ME results left on the device
            assert r == 0, lib.svt_amd_last_error()
            r = lib.svt_amd_encode_tu_batch(ctx, 1, 16, u16.data_ptr(), frames.data_ptr(), W, rec.data_ptr(), W, quant.data_ptr(),
                                            nz.data_ptr(), B * len(g16))
            assert r == 0, lib.svt_amd_last_error()
            if u8 is not None:
                r = lib.svt_amd_encode_tu_batch(ctx, 1, 8, u8.data_ptr(), frames.data_ptr(), W, rec.data_ptr(), W,
                                                quant.data_ptr() + 2 * 256 * B * len(g16),   # levels are stored unit after unit
                                                nz.data_ptr() + 4 * B * len(g16), B * len(g8))
                assert r == 0, lib.svt_amd_last_error()
            state["n0"] += B

        r = lib.svt_amd_picture_upload_device(ctx, 0, C.c_void_p(frames[0].data_ptr()), W, W, H)
        assert r == 0, lib.svt_amd_last_error()
        return ctx, step, (frames, rec, quant, nz)

    lanes = [make_lane(i) for i in range(max(1, a.streams))]
    ctx = lanes[0][0]
    tick = {"i": 0}

    def step():
        lanes[tick["i"] % len(lanes)][1]()
        tick["i"] += 1

    def sync_all():
        for c, _, _ in lanes:
            lib.svt_amd_synchronize(c)

    for _ in range(a.warmup):
        step()
    sync_all()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    lib.svt_amd_timer_begin(ctx)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = C.c_float()
    lib.svt_amd_timer_end(ctx, C.byref(ev_ms))
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    me_ms, me_n, prep_ms, prep_n = C.c_float(), C.c_int(), C.c_float(), C.c_int()
    lib.svt_amd_kernel_time(ctx, b"me_search", C.byref(me_ms), C.byref(me_n))
    lib.svt_amd_kernel_time(ctx, b"prep", C.byref(prep_ms), C.byref(prep_n))
    ois_ms, ois_n = C.c_float(), C.c_int()
    lib.svt_amd_kernel_time(ctx, b"ois", C.byref(ois_ms), C.byref(ois_n))

    if rank == 0:
        pictures = world * B * a.steps
        fps = pictures / dt
        nlcu = S.lcu_count(W, H)
        # algorithmic HBM bytes of one ME launch: source + 1 reference, each
        # full + 1/4 + 1/16 planes (1.3125 bytes/pel), plus the per-LCU result records
        algo_bytes = B * (2 * 1.3125 * W * H + nlcu * C.sizeof(S.MeLcuResult))  # one launch = B pictures
        achieved = algo_bytes / (me_ms.value * 1e-3) / 1e9 if me_ms.value > 0 else 0.0
        res = {
            "metric": "encoded fps (hot path: picture prep + motion estimation + open-loop intra search + residual DCT/quantiser/"
                      "reconstruction)", "value": round(fps, 2),
            "unit": "fps", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1080p 8-bit encMode 9 low-delay P (BASELINE configs[1]): per picture pad+decimate+"
                                   "half-pel planes, open-loop ME (HME L0/L1, full-pel 85 PU, sub-pel) of 510 LCUs "
                                   "vs previous picture, open-loop intra search, and the fused encode-pass unit (residual, DCT, quantiser, inverse "
                                   "quantiser, inverse DCT, reconstruction) over the luma plane as 16x16 transform units; mode decision "
                                   "control flow stays on the host",
                       "width": W, "height": H, "pictures_per_step_per_gpu": B, "mpix_per_s": round(fps * W * H / 1e6, 1),
                       "parallelism": "pictures sharded over ranks, no collective", "streams_per_gpu": len(lanes)},
            "roofline": {"bound": "hbm", "kernel": "k_me<0> (hme) + k_me<1> (search), one batch = 2 launches", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         # rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE of the two ME kernels per batch at this configuration
                         # (profiles/r01_g_pmc_fetch.csv, _write.csv; separate passes, KB -> bytes, uncorrected)
                         "traffic": int((134699.6 + 85340.2 + 224357.7 + 38430.8) * 1024) if B == 16 else None,
                         "algorithmic_bytes_per_launch": int(algo_bytes), "avg_launch_ms": round(me_ms.value, 4),
                         "launches_timed": me_n.value, "pictures_per_launch": B,
                         "prep_avg_ms": round(prep_ms.value, 4), "prep_launches": prep_n.value,
                         "ois_avg_launch_ms": round(ois_ms.value, 4), "ois_launches": ois_n.value,
                         "event_ms_total": round(ev_ms.value, 3)},
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_port(params, oparams)
            ref = reference_encoder_fps()
            if ref:
                res["reference_encoder"] = ref
        print(json.dumps(res), flush=True)

    for c, _, _ in lanes:
        lib.svt_amd_context_destroy(c)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
