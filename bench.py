#!/usr/bin/env python3
"""bench.py - throughput of the MI355X-native SVT-HEVC block-analysis hot path on BASELINE.json's headline configuration.

Workload (default, `--config 2`): BASELINE configs[2] = 3840x2160 8-bit, encMode 7, random access with 2 hierarchical levels
(B pictures, two reference lists, HME level-0 64x32 + level 1, full-pel 16x9, selective sub-pel), controls exactly as the
reference encoder derived them for this configuration (tests/golden/me_b_3840x2160_m7.npz, ois_ib_3840x2160_m7.npz).
`--config 1` = BASELINE configs[1] (1920x1080 encMode 9 low-delay P, one list).

One "step" = one pass of the front half of the hot path over --lanes (default 2) batches of B pictures, THROUGH THE HOST BOUNDARY:
    pinned host luma --H2D--> picture preparation (pad + 1/4 + 1/16 decimation + half-pel planes, one launch)
    -> open-loop motion estimation of every LCU against both references (HME, full-pel 85-PU search, half/quarter-pel,
       bi-prediction, candidate records: 2 kernels per list, one batched launch each)
    -> open-loop intra search (one batched launch, reads the ME results left in HBM)
    --device pack--> the compact wire records the reference side reads (MeCuResults x 85 = 2,040 B/LCU; the <= 9 OIS candidates a
       P / B picture writes per CU = 3,148 B/LCU; the full 3,420 + 6,208 B records stay in HBM for parity tests)
    --D2H--> into pinned host memory.
Three lanes (streams over the same picture slots, svt_amd_context_fork) form the pipeline: copy-in, compute, copy-out; the batches'
buffer sets rotate through them under lane events, so the copies of batches s-1 / s+1 run under the kernels of batch s and kernels
of different batches never share the GPU (the per-kernel times are stand-alone durations).  Both PCIe directions are INSIDE the
timed region; `hbm_resident_fps` is the compute lane alone.

Prints ONE JSON line (rank 0).

`metric` / `value` = BASELINE.json's metric: ENCODED fps of the whole encoder with the HIP path bound in (integration/_build/
SvtHevcEncApp_hip = the drop-in libSvtHevcEnc.so.1 + the reference's sample application) on BASELINE configs[2]'s command line, WITH THE CLOSED
LOOP ON THE DEVICE (CLOSED_LOOP_ENV below: SVT_HOOK_MD=pb - motion estimation and open-loop intra search of every picture, mode decision, merge / skip
decisions and encode pass of every open-loop P / B picture, one call per picture; `encoder_fps.coverage` says how many pictures that was), the
application's own "Average Speed" (SURVEY 8d), GATED on the bitstream being md5-identical to the unmodified reference's
(oracle/_ref/SvtHevcEncApp_ref) on the same clip in every run: a differing bitstream reports value 0.  `value` is the MEDIAN of RUNS = 3
encodes (`encoder_fps.hip.fps_runs` holds all three).  A "step" of that number = 8 encoded pictures (two mini-GOPs of the 3-layer
random-access structure): --steps K encodes 8 K pictures (K = 20: 160 >= the 150 SURVEY 8d asks for); start-up is outside the clock (the
application preloads the clip, -nb, and starts its clock after EbInitEncoder).  `cpu_baseline.value` = the unmodified reference encoder (AVX2
tables) on the same clip at ITS best threading on these hosts (-lp 32, median of 3; the default-threading run and a bounded `-lp 1` run are
beside it); `vs_baseline` stays null (BASELINE.md holds no published number for this metric), `vs_cpu_baseline` = value / cpu_baseline.value.
`front_half` keeps the hot-path throughput of the device front half (what `value` was in rounds 1-2) and `roofline` its dominant
kernels; both are measured with --steps K batches-pairs of the loop described above, bracketed by barrier + synchronize.
N > 1: one hooked encoder per rank on its own GPU (SVT_AMD_DEVICE = local rank, host threads divided), all started behind a barrier;
value = all ranks' pictures / the slowest rank's encode time ("replicas", SURVEY 8e picture-level row; no data-path collective).
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

# one hardware queue per lane: with HIP's default of 4 the copy-out lane shares a queue with the compute lane and its stream
# markers serialise the result copies with the next batch's kernels (profiles/r02_k_timeline.txt; the library sets the same
# default for C hosts, svt-hevc_amd/csrc/context.hip).  Must be in the environment before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import svtlib as S  # noqa: E402
from golden_util import load_case  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec

CONFIGS = {
    2: dict(w=3840, h=2160, me="b_3840x2160_m7", me_pic=0, ois="ib_3840x2160_m7", ois_pic=4, enc="cfg3",
            name="4K 8-bit encMode 7 random-access B pictures (BASELINE configs[2])"),
    1: dict(w=1920, h=1080, me="p_1920x1080_m9", me_pic=0, ois="ip_1920x1080_m9", ois_pic=1, enc="cfg2",
            name="1080p 8-bit encMode 9 low-delay P pictures (BASELINE configs[1])"),
}


def synth_frames(n, w, h, seed, device):
    """Moving-texture luma frames generated on the device (uint8 [n, h, w]) - SURVEY 8d's clip."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    noise = torch.randint(0, 256, (h + 2 * n + 8, w + 4 * n + 8), dtype=torch.uint8, device=device, generator=g)
    x = torch.arange(w, device=device, dtype=torch.float32)[None, :]
    y = torch.arange(h, device=device, dtype=torch.float32)[:, None]
    frames = torch.empty((n, h, w), dtype=torch.uint8, device=device)
    for t in range(n):
        l = 128 + 50 * (torch.sin((x + 3 * t) / 17.0) + torch.cos((y + 2 * t) / 23.0)) + (noise[t:t + h, 2 * t:2 * t + w] >> 4).float()
        frames[t] = l.clamp(0, 255).to(torch.uint8)
    return frames


def cpu_baseline_reference(cfg, unique=8, frames=48):
    """The REFERENCE's own front half on this host: oracle/_ref/SvtHevcEncApp_ref (-asm 1 = AVX2 tables) encodes a bounded clip
    of the same configuration while oracle/ref_harness_front_time.c sums the thread CPU time spent inside MotionEstimateLcu and
    OpenLoopIntraSearchLcu.  value = pictures per CPU-second of those two functions = the one-core rate of the same stages."""
    import encoder_fps as E
    w, h, depth, args = E.CONFIGS[cfg["enc"]]
    with tempfile.TemporaryDirectory() as td:
        yuv, rep = os.path.join(td, "c.yuv"), os.path.join(td, "front.json")
        S.write_clip(yuv, "motion", w, h, unique, 7)
        t0 = time.perf_counter()
        E.run_app(S.REF_APP, yuv, w, h, frames, args + ["-asm", "1"], os.path.join(td, "o.265"), env={"SVT_REF_FRONT_TIME": rep},
                  nb=unique)
        wall = time.perf_counter() - t0
        r = json.load(open(rep))
    nl = S.lcu_count(w, h)
    cpu_s = (r["me_ns"] + r["ois_ns"]) * 1e-9
    pictures = r["ois_calls"] / nl
    return {"value": round(pictures / cpu_s, 3), "unit": "fps", "cores": 1, "kind": "reference",
            "sample": "%d %dx%d pictures of the same configuration (%d unique, looped) through oracle/_ref/SvtHevcEncApp_ref -asm 1 "
                      "(reference compiled in place, AVX2 tables); thread CPU time inside MotionEstimateLcu (%d LCU calls, %.2f s) + "
                      "OpenLoopIntraSearchLcu (%d LCU calls, %.2f s) summed over all threads = %.2f CPU-s; picture preparation "
                      "(decimation / padding) is outside those two functions and not counted; encoder wall %.1f s" %
                      (int(pictures), w, h, unique, r["me_calls"], r["me_ns"] * 1e-9, r["ois_calls"], r["ois_ns"] * 1e-9, cpu_s, wall),
            "host_threads": os.cpu_count(),
            "all_cores_upper_bound": round(pictures / cpu_s * (os.cpu_count() or 1), 1)}


HIP_LP = 32   # logical processors the hooked encoder is run with (see encoded_fps_leg)
# The configuration `value` is measured on: the closed loop on the device.  SVT_HOOK_MD=pb: mode decision + merge / skip decisions + encode pass of every open-loop P / B
# picture (temporal layers 1 and 2 of BASELINE configs[2] = 3 of every 4 pictures) as ONE device call per picture, on top of the front half (motion estimation + open-loop
# intra search of every picture).  The I pictures (one per second of video: 3 of 160) and the base-layer B pictures stay with the reference's code: a 4K closed-loop I
# picture takes the device 0.45 s along its wavefront (35 intra candidates x 84 units per LCU; profiles/r05_c_md_bench_i_4k_stages.json) where 32 host threads take 0.06 s,
# and everything waits for it - SVT_HOOK_MD=1 (I pictures on the device too) is measured beside it (`other_configurations`).  SVT_HOOK_PCS_POOL: PictureControlSet_t
# objects of the encoder's EncDec pool (integration/svt_hook_encdec.c; the reference sizes it MAX(4, lp / 6) for host latencies) - DESIGN 5 has the sweep.
CLOSED_LOOP_ENV = {"SVT_HOOK_MD": "pb", "SVT_HOOK_PCS_POOL": "16", "SVT_HOOK_EP_LANES": "12", "SVT_HOOK_FRONT_LANES": "4"}
RUNS = 3          # `value` and the CPU baseline beside it are medians of this many encodes (84 vs 102 fps between boxes, and run to run on one)


class FileSync:
    """Barrier / all-gather of short strings between the ranks of ONE node through files - what the encoder leg needs (the reference's md5 to every rank, a start barrier, the
    slowest rank's time) BEFORE any rank has opened its GPU: the leg's child encoders must not share the part's hardware queues with this process's streams (see main), and a
    process group on the nccl backend opens the device.  Keyed by the launch (MASTER_PORT + the launcher's pid: all ranks of one torch.distributed.run share both)."""

    def __init__(self, rank, world, root=None):
        self.rank, self.world, self.n = rank, world, 0
        key = "svtbench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("SVT_BENCH_SYNC_KEY", str(os.getppid())))
        self.dir = os.path.join(root or tempfile.gettempdir(), key)
        os.makedirs(self.dir, exist_ok=True)

    def exchange(self, value, timeout_s=1800.0):
        """every rank gives a string, every rank gets all of them in rank order; returns only when every rank has given its own (a barrier)"""
        self.n += 1
        path = lambda r: os.path.join(self.dir, "x%d_r%d" % (self.n, r))  # noqa: E731
        with open(path(self.rank) + ".tmp", "w") as f:
            f.write(str(value))
        os.replace(path(self.rank) + ".tmp", path(self.rank))
        t0 = time.monotonic()
        while not all(os.path.exists(path(r)) for r in range(self.world)):
            if time.monotonic() - t0 > timeout_s:
                raise RuntimeError("FileSync: rank %d waited %.0f s for the other ranks at step %d (%s)" % (self.rank, timeout_s, self.n, self.dir))
            time.sleep(0.02)
        return [open(path(r)).read() for r in range(self.world)]

    def close(self, timeout_s=600.0):
        """rank 0 removes the directory - after every other rank has said that it will not look at it again"""
        self.exchange("bye")
        done = lambda r: os.path.join(self.dir, "done_r%d" % r)  # noqa: E731
        if self.rank != 0:
            open(done(self.rank), "w").close()
            return
        t0 = time.monotonic()
        while not all(os.path.exists(done(r)) for r in range(1, self.world)) and time.monotonic() - t0 < timeout_s:
            time.sleep(0.02)
        shutil.rmtree(self.dir, ignore_errors=True)


class GpuBusy:
    """gpu_busy_percent of the device (amdgpu sysfs) sampled every 50 ms while an encoder child runs"""

    def __init__(self, index):
        import threading
        # every card of the box is sampled: which sysfs card is the visible device differs between boxes (card0 read 0 % through a whole encode on one of them); the
        # summary reports the busiest one
        self.paths = sorted(glob.glob("/sys/class/drm/card*/device/gpu_busy_percent"))
        self.path = self.paths[0] if self.paths else None
        self.samples, self.stop = {p: [] for p in self.paths}, threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self.stop.is_set():
            for p in self.paths:
                try:
                    self.samples[p].append(int(open(p).read().strip()))
                except Exception:  # noqa: BLE001
                    pass
            self.stop.wait(0.05)

    def __enter__(self):
        if self.path:
            self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        if self.path:
            self.thread.join(timeout=1.0)

    def summary(self):
        best = None
        for p, v in self.samples.items():
            if v and (best is None or np.mean(v) > np.mean(self.samples[best])):
                best = p
        if best is None:
            return None
        a = np.array(self.samples[best], np.float64)
        return {"mean": round(float(a.mean()), 1), "max": float(a.max()), "samples": len(a), "source": best, "cards_sampled": len(self.paths)}


def _median_run(runs):
    """the run with the median fps of a list of run_app records (+ the spread)"""
    ok = sorted((r for r in runs if r.get("fps")), key=lambda r: r["fps"])
    if not ok:
        return runs[0] if runs else {"fps": None, "md5": None}
    m = dict(ok[len(ok) // 2])
    m["fps_runs"] = [r["fps"] for r in runs if r.get("fps")]
    m["fps_min"], m["fps_max"] = ok[0]["fps"], ok[-1]["fps"]
    return m


def _coverage(report_path):
    import re
    lines = [l.strip() for l in open(report_path) if "mode decision" in l] if os.path.exists(report_path) else []
    m = re.search(r"mode decision: (\d+) pictures \((\d+) of them P / B; (\d+) LCUs\).*?; (\d+) pictures outside", " ".join(lines))
    cover = {"pictures_on_device": int(m.group(1)), "p_b_pictures_on_device": int(m.group(2)), "pictures_left_to_the_reference_code": int(m.group(4))} if m else None
    return cover, lines


def encoded_fps_leg(enc_cfg, steps, rank, local_rank, world, sync):
    """The BASELINE metric, on the CLOSED LOOP (CLOSED_LOOP_ENV).  Rank 0 first encodes the clip with the unmodified reference (its md5 is the gate); then every rank runs
    the hooked encoder on its own GPU behind a barrier, RUNS times at N = 1.  Returns (rank 0) the record of the JSON line's `encoder_fps` with `value` = pictures of all
    ranks / slowest rank's encode time of the MEDIAN run, 0 when any bitstream of any run differs.  Beside it (N = 1): the reference at the same -lp (its best threading;
    median of RUNS - what cpu_baseline leads with), the hooked encoder with only the front half on the device (the configuration `value` was measured on up to round 4),
    with the I pictures left to the host (SVT_HOOK_MD=pb), and at default threading."""
    import encoder_fps as E
    frames, unique = 8 * max(1, steps), 16
    w, h, depth, args = E.CONFIGS[enc_cfg]
    args = list(args) + ["-asm", "1"]
    user = {k: v for k, v in os.environ.items() if k.startswith("SVT_HOOK_") or k.startswith("SVT_AMD_")}
    closed = dict(CLOSED_LOOP_ENV, **user)
    out = {"config": enc_cfg, "frames": frames, "unique_frames": unique, "args": " ".join(args), "host_threads": os.cpu_count(), "switches": closed, "runs": RUNS if world == 1 else 1}
    td = tempfile.mkdtemp(prefix="svtenc_r%d_" % rank, dir="/tmp")
    try:
        yuv = os.path.join(td, "clip.yuv")
        if depth == 10:
            S.write_clip10_compressed(yuv, "motion", w, h, unique, 7)
        else:
            S.write_clip(yuv, "motion", w, h, unique, 7)
        ref_md5 = ""
        if rank == 0:
            try:
                ref = E.run_app(S.REF_APP, yuv, w, h, frames, args, os.path.join(td, "ref.265"), nb=unique)
                out["reference"], ref_md5 = ref, ref["md5"]
            except Exception as e:
                out["reference"] = {"error": str(e)[-300:]}
        if world > 1:
            ref_md5 = sync.exchange(ref_md5 if rank == 0 else "")[0]      # rank 0's reference md5 to every rank; every rank's clip is written: the encoders start together
        env = dict(closed, SVT_AMD_DEVICE=str(local_rank))
        hargs = list(args)
        # host threads of the encoder with the HIP path bound in (-lp, the application's own switch; the bitstream does not depend on it): about 32 logical processors is
        # where the reference's own pipeline is fastest on these hosts (profiles/r03_ab_fps_threads.txt: reference 46 / 70 / 78 / 71 / 61 / 55 / 51 fps at 8 / 16 / 32 / 48 /
        # 64 / 128 / 256) - `cpu_baseline` leads with the reference AT THAT -lp - and N ranks share the host's threads anyway.
        ncpu = os.cpu_count() or 1
        lp = max(1, ncpu // world) if world > 1 else ncpu
        lp = min(lp, HIP_LP)
        if lp < ncpu:
            hargs += ["-lp", str(lp)]
        out["hip_threads"] = lp
        rp = os.path.join(td, "report.txt")
        hips, busy = [], None
        # SVT_HOOK_WATCHDOG: a wedged encoder ends itself after 20 s without an LCU through EncodePass and says what every picture object and the launch budget held
        # (integration/svt_hook_encdec.c).  A run that aborts or times out is NEVER repeated: it is reported (`aborted_runs`), the line says `unstable` and `value` is 0 -
        # an intermittent wedge of the shipped configuration must not hide behind a retry
        env.setdefault("SVT_HOOK_WATCHDOG", "20")
        aborted = []
        with GpuBusy(local_rank) as gb:
            for k in range(RUNS if world == 1 else 1):
                try:
                    hips.append(E.run_app(E.HIP_APP, yuv, w, h, frames, hargs, os.path.join(td, "hip.265"), env=dict(env, **({"SVT_HOOK_REPORT": rp} if k == 0 else {})), nb=unique,
                                          timeout=240))
                except Exception as e:
                    aborted.append(str(e)[-1200:])
                    hips.append({"error": str(e)[-300:], "fps": None, "md5": None})
            busy = gb.summary()
        out["aborted_runs"] = aborted
        out["unstable"] = bool(aborted)
        hip = _median_run(hips)
        out["gpu_busy"] = dict(busy, what="amdgpu gpu_busy_percent sampled every 50 ms over the %d closed-loop encodes (start-up and clip preload included)" % len(hips)) if busy else None
        out["coverage"], out["report"] = _coverage(rp)
        if rank == 0 and world == 1:
            try:
                if lp < ncpu:
                    r2 = _median_run([E.run_app(S.REF_APP, yuv, w, h, frames, hargs, os.path.join(td, "ref_lp.265"), nb=unique) for _ in range(RUNS)])
                    out["reference_same_threads"] = dict(r2, args="-lp %d" % lp, same_bitstream_as_default_threading=r2["md5"] == ref_md5)
                # the other configurations of the same encode, one run each
                side = {}
                for tag, e2, a2 in (("front_half_only", dict(env, SVT_HOOK_MD="off", SVT_HOOK_PCS_POOL="off"), hargs),
                                    ("front_half_only_with_the_pool_of_the_closed_loop", dict(env, SVT_HOOK_MD="off"), hargs),
                                    ("closed_loop_i_pictures_on_device", dict(env, SVT_HOOK_MD="1"), hargs),
                                    ("closed_loop_default_threads", env, list(args))):
                    try:
                        r = E.run_app(E.HIP_APP, yuv, w, h, frames, a2, os.path.join(td, "side.265"), env=e2, nb=unique)
                        side[tag] = {"fps": r["fps"], "bitstream_identical": r["md5"] == ref_md5}
                    except Exception as e:
                        side[tag] = {"error": str(e)[-300:]}
                side["front_half_only_with_the_pool_of_the_closed_loop"]["what"] = ("the pool control (VERDICT r5): the reference's OWN EncDec on the host threads (SVT_HOOK_MD=off) with the "
                                                                                    "EncDec pool raised to the closed loop's 16 picture control sets - against front_half_only (the reference's "
                                                                                    "5 objects) it says what the pool alone buys a host-side EncDec")
                side["front_half_only"]["what"] = "SVT_HOOK_MD=off: motion estimation + open-loop intra search on the device, the reference's own EncDec on the host threads (`value` of rounds 3-4)"
                side["closed_loop_i_pictures_on_device"]["what"] = "SVT_HOOK_MD=1: the I pictures decided and encoded by the device call as well (a 4K closed-loop I picture takes it ~0.45 s along its wavefront)"
                out["other_configurations"] = side
            except Exception as e:
                out["reference_same_threads"] = {"error": str(e)[-300:]}
        ok = all(bool(r.get("md5")) and r["md5"] == ref_md5 for r in hips)
        secs = frames / hip["fps"] if hip.get("fps") else float("inf")
        if world > 1:
            secs = max(float(v) for v in sync.exchange(repr(secs if ok else float("inf"))))     # the slowest rank; inf when any rank's bitstream differs
        elif not ok:
            secs = float("inf")
        out["hip"] = hip
        out["bitstream_identical"] = secs != float("inf")
        out["value"] = round(world * frames / secs, 2) if secs != float("inf") else 0.0
        if out.get("reference", {}).get("fps") and hip.get("fps"):
            out["hip_over_reference"] = round(hip["fps"] / out["reference"]["fps"], 3)               # reference: default threading
            if out.get("reference_same_threads", {}).get("fps"):
                out["hip_over_reference_same_threads"] = round(hip["fps"] / out["reference_same_threads"]["fps"], 3)
        return out
    finally:
        shutil.rmtree(td, ignore_errors=True)


def md_kernel_leg(w, h):
    """`roofline_md`: the mode-decision kernel (k_md_picture) and the encode-pass kernel behind it (k_encode_picture) on recorded pictures of BASELINE configs[2] - the unmodified reference (oracle/_ref, prebuilt)
    encodes 5 pictures here with the recording harness on; the device decides + encodes its three open-loop B pictures from the recorded inputs (decisions compared with the
    reference's, leaf for leaf), timed by HIP events on the call's stream; a second pass collects the stage clocks of the LCU chain."""
    import md_bench
    g = md_bench.record_inter(w, h, 7, frames=5, kind="motion", levels=2, ref=None)
    lib = S.load_product()
    r = md_bench.run_inter(lib, g, reps=4, encode=True, check=True, profile=False)
    st = md_bench.run_inter(lib, g, reps=2, encode=True, check=False, profile=True)
    per = r["kernel"]["ms_by_hip_events"]
    nlcu = r["lcus"]
    # SURVEY 8d, EncDec: source 1.5 B/pel + two reference pictures 2 x 1.5 + reconstruction written 1.5 + mode / vector / coefficient records ~ 3 B/pel = 9 B/pel
    algo = 9.0 * w * h
    # round 6: the picture-level call is TWO kernels on one stream - k_md_picture (the decisions: the latency-bound wavefront) and k_encode_picture behind it (the encode
    # pass of every LCU, as wide as the device).  The line keeps round 5's unit: one picture = both kernels, their durations added, against the same 9 B/pel
    for p_ in per:
        p_["md_plus_encode_pass_ms"] = round(p_["kernel_ms"] + (p_.get("encode_pass_kernel_ms") or 0.0), 3)
    worst = max(p["md_plus_encode_pass_ms"] for p in per)
    avg = sum(p["md_plus_encode_pass_ms"] for p in per) / len(per)
    out = {"bound": "hbm", "kernel": "k_md_picture<true> (ModeDecisionLcu of every LCU of a picture: ONE launch, wavefront on the device) + k_encode_picture behind it "
                                     "(EncodePass of every LCU from the work records the first left in HBM)",
           "workgroups": r["kernel"]["workgroups"], "lds_bytes_per_workgroup": int(lib.svt_amd_debug_md_kernel_lds_bytes(1, 1)), "waves_per_cu": 4,
           "pictures": per, "algorithmic_bytes_per_launch": int(algo),
           "avg_launch_ms": round(avg, 3), "avg_md_kernel_ms": round(sum(p["kernel_ms"] for p in per) / len(per), 3),
           "avg_encode_pass_kernel_ms": round(sum((p.get("encode_pass_kernel_ms") or 0.0) for p in per) / len(per), 3),
           "achieved": round(algo / (avg * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "traffic": None, "latency_bound": "a picture's critical path = %d wavefront steps (W/64 + 2 (H/64 - 1)) x the mode-decision time of one LCU" %
                                             ((w + 63) // 64 + 2 * ((h + 63) // 64 - 1)),
           "stage_clocks_per_lcu_on_the_chain": st["stage_clocks_per_lcu"], "decisions": "identical to the reference's ModeDecisionLcu records (checked in this run)",
           "worst_picture_ms": worst, "lcus": nlcu}
    out["frac"] = round(out["achieved"] / HBM_PEAK_GBS, 7)
    return out


def md_kernel_pmc():
    """HBM traffic of k_md_picture + k_encode_picture (one picture), measured: tools/md_bench.py (the three open-loop B pictures of a 5-picture 4K encode, one call each + warm-up) under rocprofv3
    --pmc FETCH_SIZE and, in a separate pass, WRITE_SIZE (the TCC block cannot hold both; KiB units; FETCH_SIZE doubled on gfx950 - MI355X_MICROARCH.md "HBM"), per
    launch; the same rows carry the kernel's register / LDS / private-segment sizes."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    out, meta = {}, None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        td = tempfile.mkdtemp(prefix="svtpmc_md_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp", MD_BENCH_PICTURES="all", MD_BENCH_CLIP="motion")
            r = subprocess.run([rocprof, "--pmc", counter, "--output-format", "csv", "-d", td, "--", sys.executable, os.path.join(ROOT, "tools", "md_bench.py"),
                                "3840", "2160", "7", "1", "inter", "5"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
            files = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed: %s" % (counter, (r.stderr or r.stdout)[-300:])
            total, n = 0.0, 0
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != counter:
                        continue
                    if "k_md_picture" in row["Kernel_Name"]:
                        total += float(row["Counter_Value"])
                        n += 1
                        meta = {"vgpr": int(row["VGPR_Count"]), "agpr": int(row.get("Accum_VGPR_Count") or 0), "sgpr": int(row["SGPR_Count"]),
                                "static_lds_bytes": int(row["LDS_Block_Size"]), "scratch_bytes_per_lane": int(row["Scratch_Size"])}
                    elif "k_encode_picture" in row["Kernel_Name"]:      # the encode pass of the same picture: the traffic of one picture is both kernels'
                        total += float(row["Counter_Value"])
            if not n:
                return None, "no k_md_picture dispatch in the %s pass" % counter
            out[counter] = total * 1024.0 / n
            out[counter + "_dispatches"] = n
        finally:
            shutil.rmtree(td, ignore_errors=True)
    return dict({"fetch_bytes_raw": int(out["FETCH_SIZE"]), "write_bytes_raw": int(out["WRITE_SIZE"]), "fetch_bytes_corrected": int(2 * out["FETCH_SIZE"]),
                 "per": "picture = one k_md_picture launch + the k_encode_picture launch behind it (mean over the %d pictures of tools/md_bench.py: layer-1 and layer-2 B "
                        "pictures); register / LDS / private-segment sizes: k_md_picture" % out["FETCH_SIZE_dispatches"]},
                **(meta or {})), None


def cpu_baseline_encoder(cfg, enc):
    """The reference encoder itself on this host's cores (AVX2 tables): the same clip and command line as `value` (timed by
    encoded_fps_leg), plus a bounded single-thread run (-lp 1) for the per-core figure SURVEY 8d asks for."""
    import encoder_fps as E
    if not enc or "reference" not in enc or not enc["reference"].get("fps"):
        raise RuntimeError("reference encoder run missing")
    w, h, depth, args = E.CONFIGS[cfg["enc"]]
    best = enc.get("reference_same_threads", {})
    sample = ("%d %dx%d pictures (%d unique, looped), oracle/_ref/SvtHevcEncApp_ref -asm 1 (the reference compiled in place, AVX2 tables), same command line as `value`: %s" %
              (enc["frames"], w, h, enc["unique_frames"], enc["args"]))
    if best.get("fps"):   # the reference at its best threading on these hosts (the -lp the hooked encoder runs with): the fair pair of `value`
        out = {"value": best["fps"], "unit": "fps", "cores": enc["hip_threads"], "kind": "reference", "runs": best.get("fps_runs"),
               "sample": sample + " " + best["args"] + ", median of %d encodes" % len(best.get("fps_runs") or [1]),
               "default_threading": {"fps": enc["reference"]["fps"], "cores": os.cpu_count(),
                                     "what": "the same run without -lp (all %d host threads): slower than with 32 - the reference's pipeline loses to its own thread count "
                                             "(profiles/r03_ab_fps_threads.txt)" % os.cpu_count()}}
    else:
        out = {"value": enc["reference"]["fps"], "unit": "fps", "cores": os.cpu_count(), "kind": "reference", "sample": sample + ", default threading"}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        yuv = os.path.join(td, "c.yuv")
        (S.write_clip10_compressed if depth == 10 else S.write_clip)(yuv, "motion", w, h, 5, 7)
        t0 = time.perf_counter()
        r = E.run_app(S.REF_APP, yuv, w, h, 5, list(args) + ["-asm", "1", "-lp", "1"], os.path.join(td, "o.265"), nb=5)
        out["lp1"] = {"fps": r["fps"], "pictures": 5, "wall_s": round(time.perf_counter() - t0, 1), "args": "-lp 1"}
    return out


def pmc_traffic(argv_inner, kernel_tag="k_me"):
    """HBM traffic of the dominant kernels, measured: re-runs this command (2 steps) under rocprofv3 --pmc FETCH_SIZE and, in a
    separate pass, --pmc WRITE_SIZE (the TCC block cannot hold both, MI355X_MICROARCH.md "rocprofv3 PMC slots"), sums the
    counters over the dispatches of the two ME kernels and divides by the ME batches launched.  Counter unit is KiB; on gfx950
    FETCH_SIZE tallies 128-B requests at 64 B, so the corrected figure doubles it (same guide, "HBM")."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    out = {}
    batches = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        td = tempfile.mkdtemp(prefix="svtpmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run([rocprof, "--pmc", counter, "--output-format", "csv", "-d", td, "--", sys.executable,
                                os.path.abspath(__file__)] + argv_inner, capture_output=True, text=True, timeout=600, env=env,
                               cwd="/tmp")
            files = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed: %s" % (counter, (r.stderr or r.stdout)[-300:])
            total, n = 0.0, 0
            for f in files:
                for row in csv.DictReader(open(f)):
                    if kernel_tag in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        total += float(row["Counter_Value"])
                        n += 1
            inner = [l for l in r.stdout.splitlines() if l.startswith("{")]
            info = json.loads(inner[-1])
            batches = info["me_batches"]
            out[counter] = total * 1024.0 / batches
            out[counter + "_dispatches"] = n
        finally:
            shutil.rmtree(td, ignore_errors=True)
    return {"fetch_bytes_raw": int(out["FETCH_SIZE"]), "write_bytes_raw": int(out["WRITE_SIZE"]),
            "fetch_bytes_corrected": int(2 * out["FETCH_SIZE"]),
            "per": "ME batch (all dispatches of k_me<0> + k_me<1> of one svt_amd_me_batch_launch)",
            "dispatches_counted": out["FETCH_SIZE_dispatches"], "batches": batches}, None


def recon_exchange_leg(lib, root, rank, world, dev, reps=10, limit_s=120.0):
    """N > 1 only: the one data-path collective of the closed-loop design (svt-hevc_amd/csrc/comm.hip) - every rank owns a
    rectangle of tiles of a finished 8K 10-bit reference picture (BASELINE configs[4]: 4 tile columns) and all-gathers it over RCCL.
    Runs in a helper thread with a time limit so that a collective that does not come up can never take the bench line with it."""
    import threading
    import torch.distributed as dist
    out = {}

    def work():
        try:
            W8, H8, bps = 7680, 4320, 2
            cols, rows = 4, (2 if world == 8 else 1)

            class Rect(C.Structure):
                _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("w", C.c_uint16), ("h", C.c_uint16)]
            lib.svt_amd_tile_partition.argtypes = [C.c_uint16, C.c_uint16, C.c_int, C.c_int, C.c_int, C.POINTER(Rect), C.POINTER(C.c_int)]
            rects = (Rect * world)()
            if lib.svt_amd_tile_partition(W8, H8, cols, rows, world, rects, None) != 0:
                out["error"] = "no tile partition for %d ranks: %s" % (world, lib.svt_amd_last_error().decode())
                return
            ident = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                buf = (C.c_char * 128)()
                lib.svt_amd_comm_unique_id.argtypes = [C.c_void_p]
                assert lib.svt_amd_comm_unique_id(buf) == 0, lib.svt_amd_last_error()
                ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            ident = ident.to(dev)
            dist.broadcast(ident, 0)
            idb = (C.c_char * 128).from_buffer_copy(bytes(ident.cpu().numpy().tobytes()))
            lib.svt_amd_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            assert lib.svt_amd_comm_init(root, world, rank, idb) == 0, lib.svt_amd_last_error()
            planes = [torch.full((hh, ww), rank + 1, dtype=torch.int16, device=dev) for hh, ww in ((H8, W8), (H8 // 2, W8 // 2), (H8 // 2, W8 // 2))]
            ptrs = (C.c_void_p * 3)(*[p.data_ptr() for p in planes])
            pitch = (C.c_uint32 * 3)(W8 * bps, W8 // 2 * bps, W8 // 2 * bps)
            lib.svt_amd_recon_exchange.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_int, C.POINTER(Rect), C.c_int, C.c_int]
            torch.cuda.synchronize()
            for _ in range(2):
                assert lib.svt_amd_recon_exchange(root, ptrs, pitch, bps, rects, world, rank) == 0, lib.svt_amd_last_error()
            lib.svt_amd_synchronize(root)
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                assert lib.svt_amd_recon_exchange(root, ptrs, pitch, bps, rects, world, rank) == 0, lib.svt_amd_last_error()
            lib.svt_amd_synchronize(root)
            dist.barrier()
            ms = (time.perf_counter() - t0) / reps * 1e3
            # every rank must now hold every rank's fill value in that rank's rectangle
            ok = all(int(planes[0][r.y + r.h // 2, r.x + r.w // 2]) == i + 1 for i, r in enumerate(rects))
            total = W8 * H8 * bps * 3 // 2
            out.update({"picture": "7680x4320 10-bit 4:2:0 (BASELINE configs[4]), %d x %d tiles over %d ranks" % (cols, rows, world),
                        "bytes_per_picture": total, "ms_per_exchange": round(ms, 3), "correct": bool(ok),
                        "received_GBps_per_gpu": round(total * (world - 1) / world / (ms * 1e-3) / 1e9, 2)})
            # the closed loop composed across ranks: one 4K B picture in `world` tile columns, a rank per rectangle - encode pass of
            # the rank's LCUs, all-gather of the finished planes, padding: the next pictures' reference picture is complete on every rank
            try:
                import encodepass_bench as EPB
                tr = EPB.tile_ranks_leg(lib, root, rank, world, barrier=dist.barrier)
                secs = torch.tensor([tr.pop("seconds_per_picture")], dtype=torch.float64, device=dev)
                dist.all_reduce(secs, op=dist.ReduceOp.MAX)
                tr["ms_per_picture_slowest_rank"] = round(float(secs.item()) * 1e3, 2)
                tr["pictures_per_s"] = round(1.0 / float(secs.item()), 1)
                out["tile_ranks"] = tr
            except Exception as e:  # noqa: BLE001
                out["tile_ranks"] = {"error": str(e)[-300:]}
            lib.svt_amd_comm_destroy.argtypes = [C.c_void_p]
            lib.svt_amd_comm_destroy(root)
        except Exception as e:  # noqa: BLE001 - reported, never fatal for the bench line
            out["error"] = str(e)[-300:]

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(limit_s)
    if th.is_alive():
        return {"error": "timed out after %.0f s" % limit_s, "hung": True}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="pictures per step per GPU (default: 64 at 4K, 128 at 1080p)")
    ap.add_argument("--lanes", type=int, default=2, help="buffer sets in flight per GPU (each holds one batch); 2 <= n <= 4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encoder-fps", action="store_true")
    ap.add_argument("--no-encode-pass", action="store_true")
    ap.add_argument("--no-pmc", action="store_true")
    ap.add_argument("--inner", action="store_true", help="(internal) counter pass: GPU loop only, prints the batch count")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
    # ---- the headline first: md5-gated encoded fps of the whole encoder with the HIP path bound in.  The encoder is a child process with two dozen streams of its
    # own; measured BEFORE this process opens the device, so that the two do not share the part's hardware queues (the same leg after the loops below: 87-97 fps where the
    # stand-alone sweeps of the same binary read 103-117, profiles/r05_ah_bench_encoder_leg_last.json).  N > 1: the ranks meet through files (FileSync), not through the process
    # group - the nccl backend opens the device.
    enc = None
    if not a.inner and not a.no_encoder_fps:
        sync = FileSync(rank, world) if world > 1 else None
        enc = encoded_fps_leg(CONFIGS[a.config]["enc"], a.steps, rank, local_rank, world, sync)
        if sync:
            sync.close()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    cfg = CONFIGS[a.config]
    W, H = cfg["w"], cfg["h"]
    lib = S.load_product()  # fails loudly when the HIP library is absent
    vp, i32 = C.c_void_p, C.c_int

    def ok(rc):
        assert rc == 0, lib.svt_amd_last_error()

    B = a.batch or (64 if a.config == 2 else 128)
    g = load_case(cfg["me"])
    params = S.params_from_record(g["params"][cfg["me_pic"]])
    og = np.load(os.path.join(S.GOLDEN_DIR, "ois_%s.npz" % cfg["ois"]))
    oparams = S.ois_params_from_record(og["params"][cfg["ois_pic"]])
    assert not oparams.slice_is_intra and params.luma_width == W and oparams.luma_width == W
    nlists = params.num_lists
    nlcu = S.lcu_count(W, H)
    me_rec_b = C.sizeof(S.MeLcuResult)                       # full record in HBM (what the ME kernels write: algorithmic bytes)
    ois_nc = lib.svt_amd_ois_compact_candidates(C.byref(oparams))
    me_b, ois_b = S.ME_PU_COUNT * 24, S.ME_PU_COUNT * ois_nc * 4 + 88   # compact wire records that cross PCIe
    NL = a.lanes
    root = vp()
    ok(lib.svt_amd_context_create(local_rank, W, (H + 7) & ~7, NL * B, C.byref(root)))
    # pinned host input: B distinct frames of the moving-texture clip (shared by both lanes)
    h_in = vp()
    ok(lib.svt_amd_host_alloc(root, B * W * H, C.byref(h_in)))
    frames = synth_frames(B, W, H, 1234 + 17 * rank, dev)
    torch.cuda.synchronize()
    ok(lib.svt_amd_device_download(root, h_in, vp(frames.data_ptr()), B * W * H))
    del frames
    # three lanes = three streams over the same picture slots: copy-in (H2D), compute (every kernel), copy-out (device pack + D2H).
    # NL buffer sets (staging, B picture slots, pinned result buffers) rotate through them; lane events order the stages, the host
    # never waits inside the loop.  Kernels of different batches therefore never share the GPU: the per-kernel times below are
    # stand-alone durations, and the copies of batch s-1 / s+1 run under the kernels of batch s.
    lane_in, lane_k, lane_out = vp(), vp(), vp()
    for l in (lane_in, lane_k, lane_out):
        ok(lib.svt_amd_context_fork(root, C.byref(l)))
    sets = []
    for li in range(NL):
        d_stage, h_me, h_ois, d_me, d_ois = vp(), vp(), vp(), vp(), vp()
        ok(lib.svt_amd_device_alloc(lane_in, B * W * H, C.byref(d_stage)))
        ok(lib.svt_amd_device_alloc(lane_k, B * nlcu * me_b, C.byref(d_me)))      # packed wire records of the batch, device side
        ok(lib.svt_amd_device_alloc(lane_k, B * nlcu * ois_b, C.byref(d_ois)))
        ok(lib.svt_amd_host_alloc(lane_out, B * nlcu * me_b, C.byref(h_me)))
        ok(lib.svt_amd_host_alloc(lane_out, B * nlcu * ois_b, C.byref(h_ois)))
        slots = (i32 * B)(*[li * B + i for i in range(B)])
        ptrs = (vp * B)(*[d_stage.value + i * W * H for i in range(B)])
        jobs, ojobs = (S.MeJob * B)(), (S.OisJob * B)()
        for i in range(B):  # picture i against its neighbours in the batch (list 0 = previous, list 1 = next)
            jobs[i].params, jobs[i].cur_slot = params, slots[i]
            jobs[i].ref_slot[0] = slots[(i - 1) % B]
            jobs[i].ref_slot[1] = slots[(i + 1) % B]
            ojobs[i].params, ojobs[i].cur_slot = oparams, slots[i]
        sets.append(dict(d_stage=d_stage, h_me=h_me, h_ois=h_ois, d_me=d_me, d_ois=d_ois, slots=slots, ptrs=ptrs, jobs=jobs, ojobs=ojobs))
    lanes = [dict(ctx=lane_in), dict(ctx=lane_k), dict(ctx=lane_out)]

    counts = {"batches": 0}
    EV_STAGE, EV_READY = 0, NL   # event indices per set: [k] stage / slots handed over, [NL + k] results ready

    direct = os.environ.get("SVT_BENCH_D2H", "sdma") == "direct"   # diagnosis: pack kernels write the pinned host arrays themselves
    skip = os.environ.get("SVT_BENCH_SKIP")   # diagnosis only: "in" / "out" drops that copy stage from the pipeline
    trace = os.environ.get("SVT_BENCH_TRACE")
    if trace:  # host time inside every library call of a step (which call makes the host wait?)
        import functools
        acc = {}

        rawlib = lib

        class _Timed:
            def __getattr__(self, name):
                f = getattr(rawlib, name)

                @functools.wraps(f)
                def g(*aa):
                    t0 = time.perf_counter()
                    r = f(*aa)
                    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
                    return r
                return g
        lib = _Timed()

    def step(k, copies=True):
        """one batch through the pipeline on buffer set k"""
        L = sets[k]
        if copies:
            ok(lib.svt_amd_lane_event_wait(lane_in, lane_k, EV_STAGE + k))     # the planes of the set's previous batch are built
            if skip != "in":
                ok(lib.svt_amd_device_upload_async(lane_in, L["d_stage"], h_in, B * W * H))
            ok(lib.svt_amd_lane_event_record(lane_in, EV_STAGE + k))
            ok(lib.svt_amd_lane_event_wait(lane_k, lane_in, EV_STAGE + k))
            ok(lib.svt_amd_lane_event_wait(lane_k, lane_out, EV_STAGE + k))    # the set's previous records have left the device
        ok(lib.svt_amd_picture_upload_device_batch(lane_k, B, L["slots"], L["ptrs"], W, W, H))
        if copies:
            ok(lib.svt_amd_lane_event_record(lane_k, EV_STAGE + k))
        ok(lib.svt_amd_me_batch_launch(lane_k, L["jobs"], B))
        ok(lib.svt_amd_ois_batch_launch(lane_k, L["ojobs"], B))
        if copies and direct:
            # diagnosis: the pack kernels of the copy-out lane write straight into the pinned host arrays (no copy engine; their
            # PCIe-bound waves hold CUs the searches want, so this is slower than the default below)
            ok(lib.svt_amd_lane_event_record(lane_k, EV_READY + k))
            ok(lib.svt_amd_lane_event_wait(lane_out, lane_k, EV_READY + k))
            if skip != "out":
                ok(lib.svt_amd_records_pack_batch_async(lane_out, L["slots"], B, ois_nc, L["h_me"], L["h_ois"]))
            ok(lib.svt_amd_lane_event_record(lane_out, EV_STAGE + k))
        elif copies:
            # the compact wire records (what the reference side of the boundary reads, include/svt_hevc_amd.h): packed on the compute
            # lane into device arrays, moved by the copy engines on the copy-out lane with two copies per batch
            ok(lib.svt_amd_records_pack_batch_async(lane_k, L["slots"], B, ois_nc, L["d_me"], L["d_ois"]))
            ok(lib.svt_amd_lane_event_record(lane_k, EV_READY + k))
            ok(lib.svt_amd_lane_event_wait(lane_out, lane_k, EV_READY + k))
            if skip != "out":  # two copies for the whole batch
                ok(lib.svt_amd_device_download_async(lane_out, L["h_me"], L["d_me"], B * nlcu * me_b))
                ok(lib.svt_amd_device_download_async(lane_out, L["h_ois"], L["d_ois"], B * nlcu * ois_b))
            ok(lib.svt_amd_lane_event_record(lane_out, EV_STAGE + k))
        counts["batches"] += 1

    def sync_all():
        for L in lanes:
            ok(lib.svt_amd_synchronize(L["ctx"]))

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nsteps, copies):
        barrier()
        for L in lanes:
            lib.svt_amd_timer_begin(L["ctx"])
        t0 = time.perf_counter()
        for s in range(nsteps):
            for k in range(NL):  # NL batches per step, one per buffer set
                step(k, copies)
        sync_all()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        kt = {}
        for cls in (b"me_search", b"prep", b"ois"):
            tot, n = 0.0, 0
            for L in lanes:
                ms, k, ev = C.c_float(), C.c_int(), C.c_float()
                lib.svt_amd_kernel_time(L["ctx"], cls, C.byref(ms), C.byref(k))
                tot += ms.value * k.value
                n += k.value
            kt[cls.decode()] = (tot / n if n else 0.0, n)
        for L in lanes:
            ev = C.c_float()
            lib.svt_amd_timer_end(L["ctx"], C.byref(ev))
        return dt, kt

    for s in range(a.warmup):
        for k in range(NL):
            step(k)
    sync_all()
    dt, kt = timed(a.steps, True)

    if trace and rank == 0:
        print("host seconds per library call over %d batches: %s" % (counts["batches"], {k: round(v, 4) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}),
              file=sys.stderr, flush=True)
    if a.inner:
        if rank == 0:
            print(json.dumps({"me_batches": counts["batches"], "pictures_per_batch": B}), flush=True)
    else:
        dt_res, kt_res = timed(max(2, a.steps // 2), False)
        res_steps = max(2, a.steps // 2)

    xchg = recon_exchange_leg(lib, root, rank, world, dev) if (world > 1 and not a.inner) else None

    if rank == 0 and not a.inner:
        fps = world * NL * B * a.steps / dt
        me_ms, me_n = kt["me_search"]
        # algorithmic HBM bytes of one ME batch: per picture the source + `nlists` references, each
        # full + 1/4 + 1/16 planes (1.3125 bytes/pel), plus the per-LCU result records
        algo_bytes = B * ((1 + nlists) * 1.3125 * W * H + nlcu * me_rec_b)
        achieved = algo_bytes / (me_ms * 1e-3) / 1e9 if me_ms > 0 else 0.0
        # integer view: absolute differences of the three exhaustive stages per LCU and list against the v_sad_u8 peak
        # (4 px x 64 lanes x 4 SIMDs x 256 CUs x 2.4 GHz)
        absdiff_lcu_list = (params.hme_l0_total_w * params.hme_l0_total_h * 16 * 8 + 4 * 8 * 4 * 32 * 16 +
                            params.search_area_width * params.search_area_height * 64 * 32)
        absdiff_s = B * nlcu * nlists * absdiff_lcu_list / (me_ms * 1e-3) if me_ms > 0 else 0.0
        sad_peak = 4 * 64 * 4 * 256 * 2.4e9
        # per-kernel algorithmic rates of the other two launches of a batch
        prep_bytes = B * (W * H + 4 * (W + 136) * (H + 136) + (W // 2 + 64) * (H // 2 + 64) + (W // 4 + 32) * (H // 4 + 32))
        ois_bytes = B * (W * H + nlcu * 6208)
        value = enc["value"] if enc and "value" in enc else None
        res = {
            "metric": "encoded fps, %s, bitstream md5-identical to the unmodified reference (BASELINE.json metric; whole encoder with the "
                      "HIP path bound in, application's Average Speed)" % cfg["name"],
            "value": value if value is not None else 0.0, "unit": "fps", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(8e3 / value, 4) if value else None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "vs_cpu_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "unstable": bool(enc.get("unstable")) if enc else None,   # an encode of the measured configuration aborted (watchdog) or timed out: `value` is 0 then, never a retry's
            "fps_runs": (enc.get("hip") or {}).get("fps_runs") if enc else None,
            "peak_rss_mb_of_this_process": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1024.0, 1),
            "peak_rss_mb_of_the_encoder_processes": round(__import__("resource").getrusage(__import__("resource").RUSAGE_CHILDREN).ru_maxrss / 1024.0, 1),
            "config": {"workload": cfg["name"] + ": whole encode of %s pictures (one step = 8 pictures), command line of BASELINE.md section 2" %
                                   (enc["frames"] if enc and "frames" in enc else "n/a"),
                       "width": W, "height": H, "mpix_per_s": round(value * W * H / 1e6, 1) if value else None,
                       "switches": enc.get("switches") if enc else None,
                       "host_threads_of_the_encoder": ("-lp %d (of %d on this host); the reference's fps at the same -lp (its best) and at default threading are in "
                                                       "cpu_baseline / encoder_fps" % (enc["hip_threads"], os.cpu_count())) if enc and enc.get("hip_threads") else None,
                       "on_device": ["Decimation2D / GeneratePadding / half-pel planes (k_prep_fused)", "HME level 0 / 1, full-pel 85-PU search, sub-pel refinement, "
                                     "bi-prediction search, MeCuResults (k_me<0>, k_me<1>)", "OpenLoopIntraSearchLcu (k_ois_picture)",
                                     "ModeDecisionLcu + merge / skip decisions (AddChromaEncDec) + EncodePass of every open-loop P / B picture - temporal layers 1 and 2 "
                                     "(k_md_picture + k_encode_picture, one call per picture): %s" % (enc.get("coverage") if enc else None)] if enc else None,
                       "on_host_in_this_run": ("EncDec of the I pictures and of the base-layer P / B pictures (closed-loop intra + branch-and-depth-pillar LCUs: the reference's "
                                               "code, DESIGN 7) with their deblocking + SAO, entropy coding, picture management") if enc else None,
                       "parallelism": "one encoder per rank on its own GPU, no data-path collective" if world > 1 else "1 GPU"},
            "encoder_fps": enc,
            "front_half": {"what": "upload + picture preparation + open-loop motion estimation of %d LCUs against %d list(s) (HME L0 %dx%d + L1, "
                                   "full-pel %dx%d 85-PU search, sub-pel, bi-prediction) + open-loop intra search + result download, through the "
                                   "host boundary (PCIe inside the timed region), %d pictures per launch, %d lanes" %
                                   (nlcu, nlists, params.hme_l0_total_w, params.hme_l0_total_h, params.search_area_width,
                                    params.search_area_height, B, NL),
                           "fps": round(fps, 2), "steps": a.steps, "ms_per_step": round(dt / a.steps * 1e3, 4), "timed_region_s": round(dt, 3),
                           "pictures_per_step_per_gpu": NL * B, "hbm_resident_fps": round(world * NL * B * res_steps / dt_res, 2),
                           "h2d_bytes_per_picture": W * H, "d2h_bytes_per_picture": nlcu * (me_b + ois_b)},
            "roofline": {"bound": "hbm", "kernel": "k_me<0> (hme) + k_me<1> (search): one ME batch = %d dispatches (2 per list)" % (2 * nlists),
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": None, "algorithmic_bytes_per_launch": int(algo_bytes), "avg_launch_ms": round(me_ms, 4),
                         "launches_timed": me_n, "pictures_per_launch": B,
                         "sad_throughput": {"abs_diff_per_s": float("%.4g" % absdiff_s), "v_sad_u8_peak_per_s": sad_peak,
                                            "frac": round(absdiff_s / sad_peak, 5),
                                            "abs_diff_per_lcu_per_list": absdiff_lcu_list},
                         "other_kernels": {
                             "k_prep_fused": {"avg_ms": round(kt["prep"][0], 4), "launches": kt["prep"][1], "algorithmic_bytes": int(prep_bytes),
                                              "GBps": round(prep_bytes / (kt["prep"][0] * 1e-3) / 1e9, 1) if kt["prep"][0] > 0 else None},
                             "k_ois_picture": {"avg_ms": round(kt["ois"][0], 4), "launches": kt["ois"][1], "algorithmic_bytes": int(ois_bytes),
                                               "GBps": round(ois_bytes / (kt["ois"][0] * 1e-3) / 1e9, 1) if kt["ois"][0] > 0 else None}},
                         "avg_launch_ms_hbm_resident_loop": round(kt_res["me_search"][0], 4)},
        }
        if xchg is not None:
            res["recon_exchange"] = xchg
        if world == 1 and not a.no_pmc:
            tr, err = pmc_traffic(["--inner", "--steps", "2", "--warmup", "1", "--config", str(a.config), "--batch", str(B),
                                   "--no-cpu-baseline", "--no-encoder-fps", "--no-pmc", "--no-encode-pass"])
            if tr:
                res["roofline"]["traffic"] = tr["fetch_bytes_corrected"] + tr["write_bytes_raw"]
                res["roofline"]["traffic_detail"] = tr
            else:
                res["roofline"]["traffic_error"] = err
        if world == 1 and not a.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline_encoder(cfg, enc)
                res["cpu_baseline"]["front_half_one_core"] = cpu_baseline_reference(cfg)
            except Exception as e:  # the reference build is absent: say so, do not substitute
                res["cpu_baseline"] = {"error": str(e)[-300:]}
        if res.get("cpu_baseline", {}).get("value") and value:
            res["vs_cpu_baseline"] = round(value / res["cpu_baseline"]["value"], 3)
        if world == 1 and not a.no_encode_pass and os.path.exists(S.REF_APP):
            try:
                res["roofline_md"] = md_kernel_leg(W, H)
                if not a.no_pmc:
                    tr, err = md_kernel_pmc()
                    if tr:
                        res["roofline_md"]["traffic"] = tr["fetch_bytes_corrected"] + tr["write_bytes_raw"]
                        res["roofline_md"]["traffic_detail"] = tr
                    else:
                        res["roofline_md"]["traffic_error"] = err
            except Exception as e:
                res["roofline_md"] = {"error": str(e)[-300:]}
        if world == 1 and not a.no_encode_pass:
            # the device-resident encode pass, measured beside the front half: it is not part of `value`
            try:
                import encodepass_bench as EPB
                res["encode_pass"] = EPB.measure_b_picture(S.load_product(), root)
                # the chain `--gpus N` runs with a rank per tile rectangle (recon_exchange.tile_ranks), here with one rank owning the whole picture
                tr = EPB.tile_ranks_leg(S.load_product(), root, 0, 1)
                tr["pictures_per_s"] = round(1.0 / tr.pop("seconds_per_picture"), 1)
                res["encode_pass"]["tile_ranks_one_rank"] = tr
            except Exception as e:
                res["encode_pass"] = dict(res.get("encode_pass") or {}, error=str(e)[-300:])
        # `roofline` is the DOMINANT kernel of the run `value` comes from: with the closed loop on the device that is k_md_picture (99 % of the device time of the
        # encode, profiles/r05_d_md_timeline_pb_pool8.txt), a latency-bound wavefront kernel far from any roofline - the honest number.  The front half's ME kernels
        # (`roofline` of rounds 1-4) keep their object as `roofline_front_half`.
        if isinstance(res.get("roofline_md"), dict) and "frac" in res["roofline_md"]:
            res["roofline_front_half"] = res.pop("roofline")
            res["roofline"] = res.pop("roofline_md")
        print(json.dumps(res), flush=True)

    if xchg and xchg.get("hung"):
        os._exit(0)  # a helper thread is still inside the collective library: do not wait for it at teardown
    for L in sets:
        lib.svt_amd_device_free(lane_in, L["d_stage"])
        lib.svt_amd_device_free(lane_k, L["d_me"])
        lib.svt_amd_device_free(lane_k, L["d_ois"])
        lib.svt_amd_host_free(lane_out, L["h_me"])
        lib.svt_amd_host_free(lane_out, L["h_ois"])
    for L in lanes:
        lib.svt_amd_context_destroy(L["ctx"])
    lib.svt_amd_host_free(root, h_in)
    lib.svt_amd_context_destroy(root)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
