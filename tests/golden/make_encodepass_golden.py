#!/usr/bin/env python3
"""Generate the encode-pass (LCU) golden fixtures from the REFERENCE itself.

Runs oracle/_ref/SvtHevcEncApp_ref with SVT_REF_ENCODEPASS_DUMP set and the loop filters off (-dlf 1 -sao 0, so the reconstruction
buffer still holds the un-deblocked samples when EncodePass returns): the --wrap interposer oracle/ref_harness_encodepass_dump.c
records, for every LCU whose coding units are all intra 2Nx2N, the input contract (coding-unit list + source samples =
SvtAmdLcuWork) and what the reference's EncodePass produced (TransformUnit_t flags, quantizedCoeff, reconstruction =
SvtAmdLcuResult).  Stored as tests/golden/encodepass_<name>.npz.  Needs /root/reference.  Usage: make_encodepass_golden.py [name ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svtlib as S  # noqa: E402

CASES = {
    # name -> (clip, w, h, frames, seed, args): all-intra pictures; 416x240 has a partial right column and bottom row of LCUs
    "i_motion_416x240_m9": ("motion", 416, 240, 2, 7, ["-encMode", "9", "-intra-period", "0", "-q", "30"]),
    # worst-case residual energy at a low qp: every unit size, large levels, DC-only units rare
    "i_noise_200x136_m6": ("noise", 200, 136, 1, 11, ["-encMode", "6", "-intra-period", "0", "-q", "22"]),
    # smooth content at a high qp: many units without coefficients and many DC-only units (EncodeInvTransform's shortcut)
    "i_motion_320x192_m7_q44": ("motion", 320, 192, 1, 7, ["-encMode", "7", "-intra-period", "0", "-q", "44"]),
    # 10-bit encodes (EncodePass with is16bit: EncodeLoop16bit, 16-bit intra prediction, quantiser at qp + 12): "10" in the clip kind
    "i10_motion_416x240_m9": ("motion10", 416, 240, 1, 7, ["-encMode", "9", "-intra-period", "0", "-q", "30", "-bit-depth", "10"]),
    "i10_noise_200x136_m6": ("noise10", 200, 136, 1, 11, ["-encMode", "6", "-intra-period", "0", "-q", "24", "-bit-depth", "10"]),
    # encMode 3 / 4: the encode pass quantises with PM-core (DecoupledQuantizeInvQuantizeLoops: luma levels re-decided per 4x4 block by SSE +
    # lambda * rate; "cost" = the picture's rate tables)
    "i_motion_416x240_m4": ("motion", 416, 240, 1, 7, ["-encMode", "4", "-intra-period", "0", "-q", "30"]),
    "i10_noise_200x136_m3": ("noise10", 200, 136, 1, 11, ["-encMode", "3", "-intra-period", "0", "-q", "28", "-bit-depth", "10"]),
    "i10_motion_320x192_m7_q45": ("motion10", 320, 192, 1, 7, ["-encMode", "7", "-intra-period", "0", "-q", "45", "-bit-depth", "10"]),
    # deblocking ON (the encoder's default), SAO off: the encoder's reconstruction output is then the deblocked picture, the
    # fixture for encode pass -> boundary strengths -> deblocking chained on the device ("dlf_" prefix: `recon_*` planes are kept,
    # the per-LCU `rec_*` of the records are partly deblocked and not compared)
    "dlf_i_motion_416x240_m9": ("motion", 416, 240, 2, 7, ["-encMode", "9", "-intra-period", "0", "-q", "34"]),
    "dlf_i_noise_200x136_m6": ("noise", 200, 136, 1, 11, ["-encMode", "6", "-intra-period", "0", "-q", "38"]),
    "dlf_i10_motion_320x192_m7": ("motion10", 320, 192, 1, 7, ["-encMode", "7", "-intra-period", "0", "-q", "36", "-bit-depth", "10"]),
    # ... of P / B pictures: boundary strengths from motion vectors, reference pictures and transform-unit cbf (64x64 units: four units)
    "dlf_p_motion_416x240_m7": ("motion", 416, 240, 4, 7, ["-encMode", "7", "-pred-struct", "0", "-hierarchical-levels", "0", "-intra-period", "-1", "-q", "36"]),
    "dlf_b_motion_320x192_m6": ("motion", 320, 192, 5, 9, ["-encMode", "6", "-pred-struct", "2", "-hierarchical-levels", "2", "-intra-period", "-1", "-q", "34"]),
    # deblocking AND SAO on ("sao_" prefix): additionally every SaoGenerationDecision call of the encode (oracle/ref_harness_saodec_dump.c:
    # the statistics the reference gathered on the picture as it stood when the LCU was done, rate inputs, decided parameters); the
    # encoder's output is then the finished reconstruction - the fixture of encode pass -> deblocking -> SAO chained on the device
    "sao_i_motion_416x240_m9": ("motion", 416, 240, 2, 7, ["-encMode", "9", "-intra-period", "0", "-q", "34"]),
    "sao_i_noise_200x136_m6": ("noise", 200, 136, 1, 11, ["-encMode", "6", "-intra-period", "0", "-q", "38"]),
    "sao_b_motion_320x192_m6": ("motion", 320, 192, 5, 9, ["-encMode", "6", "-pred-struct", "2", "-hierarchical-levels", "2", "-intra-period", "-1", "-q", "34"]),
    "sao_tiles_motion_640x384_m7": ("motion", 640, 384, 2, 7, ["-encMode", "7", "-intra-period", "0", "-q", "33", "-tile_col_cnt", "2", "-tile_row_cnt", "2"]),
    # two tile columns in a random-access encode: the multi-GPU composition (a rank per tile rectangle, tests/test_tile_ranks.py) chains on it
    "sao_b_tiles_motion_512x320_m6": ("motion", 512, 320, 5, 7, ["-encMode", "6", "-pred-struct", "2", "-hierarchical-levels", "2", "-intra-period", "-1", "-q", "33",
                                                                 "-tile_col_cnt", "2", "-tile_row_cnt", "1"]),
    "sao_p_noise_320x256_m8": ("noise", 320, 256, 3, 7, ["-encMode", "8", "-pred-struct", "0", "-hierarchical-levels", "0", "-intra-period", "-1", "-q", "36"]),
    # encMode 8: allowEncDecMismatch in temporal layers > 0 - those pictures are neither deblocked nor SAO-filtered on the encoder side
    "sao_b_motion_320x192_m8": ("motion", 320, 192, 5, 9, ["-encMode", "8", "-pred-struct", "2", "-hierarchical-levels", "2", "-intra-period", "-1", "-q", "34"]),
    "sao_i10_motion_320x192_m7": ("motion10", 320, 192, 1, 7, ["-encMode", "7", "-intra-period", "0", "-q", "36", "-bit-depth", "10"]),
    # P / B pictures ("p_" / "b_" / "p10_" prefix: LCUs with inter units are recorded too, with the reference pictures they predict from and
    # the pictures' coefficient-rate tables; loop filters off as above).  Low delay P: uni-prediction, AMVP / merge / skip units, 64x64 units
    "p_motion_416x240_m7": ("motion", 416, 240, 4, 7, ["-encMode", "7", "-pred-struct", "0", "-hierarchical-levels", "0", "-intra-period", "-1", "-q", "30"]),
    # noise at a high qp: most AMVP units lose their luma coefficients in the cbf decision (EncodeTuCalcCost)
    "p_noise_200x136_m5_q48": ("noise", 200, 136, 4, 13, ["-encMode", "5", "-pred-struct", "0", "-hierarchical-levels", "0", "-intra-period", "-1", "-q", "48"]),
    # random access: bi-prediction, non-reference B pictures (the skip-cost bias of :3865-3878)
    "b_motion_320x192_m5": ("motion", 320, 192, 5, 9, ["-encMode", "5", "-pred-struct", "2", "-hierarchical-levels", "2", "-intra-period", "-1", "-q", "28"]),
    # noise: every unit carries coefficients, the luma cbf decision sees large rates
    "p_noise_200x136_m6": ("noise", 200, 136, 3, 11, ["-encMode", "6", "-pred-struct", "0", "-hierarchical-levels", "0", "-intra-period", "-1", "-q", "34"]),
    # constrained intra prediction, encMode 8: 32x32 and 64x64 AMVP units with ~200 coefficients that the luma cbf decision zeroes
    "p_noise_320x256_m8_ci": ("noise", 320, 256, 4, 7, ["-encMode", "8", "-pred-struct", "0", "-constrd-intra", "1", "-q", "40"]),
    # 10-bit random access: 16-bit bi-prediction (BiPredClipping16bit), non-reference B pictures
    "b10_motion_320x192_m6": ("motion10", 320, 192, 5, 9, ["-encMode", "6", "-pred-struct", "2", "-hierarchical-levels", "2", "-intra-period", "-1", "-q", "30",
                                                          "-bit-depth", "10"]),
    "p_motion_416x240_m3": ("motion", 416, 240, 3, 7, ["-encMode", "3", "-pred-struct", "0", "-hierarchical-levels", "0", "-intra-period", "-1", "-q", "28"]),
    "b_noise_320x192_m4": ("noise", 320, 192, 5, 13, ["-encMode", "4", "-pred-struct", "2", "-hierarchical-levels", "2", "-intra-period", "-1", "-q", "40"]),
    "p10_motion_320x192_m7": ("motion10", 320, 192, 3, 7, ["-encMode", "7", "-pred-struct", "0", "-hierarchical-levels", "0", "-intra-period", "-1", "-q", "32", "-bit-depth", "10"]),
}


def parse_dump(raw, rdt):
    """the dump of a P / B encode is a sequence of LCU records, reference-picture records and rate-table records"""
    recs, refs, costs, off = [], {}, {}, 0
    while off < len(raw):
        magic, size = np.frombuffer(raw, "<u4", 2, off)
        if magic == S.EP_MAGIC:
            assert size == rdt.itemsize, (size, rdt.itemsize)
            recs.append(np.frombuffer(raw, rdt, 1, off)[0])
        elif magic == S.EP_REF_MAGIC:
            h = np.frombuffer(raw, S.EP_REF_HEAD_DTYPE, 1, off)[0]
            sdt = np.dtype(np.uint8) if h["bps"] == 1 else np.dtype("<u2")
            rows_y = int(h["height"] + 2 * h["originY"])
            ny, nc = int(h["strideY"]) * rows_y, int(h["strideC"]) * (rows_y // 2)
            o = off + S.EP_REF_HEAD_DTYPE.itemsize
            planes = [np.frombuffer(raw, sdt, n, o + k * sdt.itemsize) for n, k in ((ny, 0), (nc, ny), (nc, ny + nc))]
            refs[int(h["poc"])] = (h, planes)
        elif magic == S.EP_COST_MAGIC:
            costs[int(np.frombuffer(raw, "<u8", 1, off + 8)[0])] = np.frombuffer(raw, np.uint8, int(size) - 16, off + 16)
        else:
            raise AssertionError("bad record magic %x at %d" % (magic, off))
        off += int(size)
    return np.array(recs, dtype=rdt), refs, costs


def run_case(name):
    kind, w, h, n, seed, args = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        yuv, dump = os.path.join(td, "clip.yuv"), os.path.join(td, "ep.dump")
        if kind.endswith("10"):
            S.write_clip10(yuv, kind[:-2], w, h, n, seed)
        else:
            S.write_clip(yuv, kind, w, h, n, seed)
        sao = name.startswith("sao_")
        dlf = name.startswith("dlf_") or sao
        rec_out, sao_dump = os.path.join(td, "rec.yuv"), os.path.join(td, "saodec.dump")
        cmd = [S.REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-asm", "0", "-sao", "1" if sao else "0",
               "-b", os.path.join(td, "out.265")] + ([] if dlf else ["-dlf", "1"]) + (["-o", rec_out] if dlf else []) + args
        env = dict(os.environ, SVT_REF_ENCODEPASS_DUMP=dump)
        if sao:
            env.update(SVT_REF_SAODEC_DUMP=sao_dump, SVT_REF_SAODEC_STRIDE="1")
        subprocess.run(cmd, env=env, check=True, stdout=subprocess.DEVNULL)
        sao_recs = None
        if sao:
            from make_saodec_golden import REC as SAO_REC
            sao_recs = np.fromfile(sao_dump, SAO_REC)
        rdt = S.EP_RECORD16_DTYPE if kind.endswith("10") else S.EP_RECORD_DTYPE
        recs, refs, costs = parse_dump(open(dump, "rb").read(), rdt)
        rec_raw = open(rec_out, "rb").read() if dlf else b""
    inter = any(t in ("p", "b", "p10", "b10") for t in name.split("_")[:2])
    assert len(recs) and (recs["record_size"] == rdt.itemsize).all(), (len(recs), rdt.itemsize)
    assert ((recs["dlf_off"] & 1) == (0 if dlf else 1)).all()   # bit 1: the LCU was not reconstructed (doRecon == 0)
    extra = {}
    if dlf:   # the encoder's reconstruction output: n frames, 4:2:0, 8-bit samples or 16-bit little-endian for 10-bit encodes
        sdt = np.dtype("<u2") if kind.endswith("10") else np.uint8
        raw = np.frombuffer(rec_raw, sdt)
        fs = w * h * 3 // 2
        assert raw.size == n * fs, (raw.size, n, fs)
        extra = {"recon_y": np.stack([raw[i * fs:i * fs + w * h].reshape(h, w) for i in range(n)]),
                 "recon_cb": np.stack([raw[i * fs + w * h:i * fs + w * h * 5 // 4].reshape(h // 2, w // 2) for i in range(n)]),
                 "recon_cr": np.stack([raw[i * fs + w * h * 5 // 4:(i + 1) * fs].reshape(h // 2, w // 2) for i in range(n)])}
    if sao:   # one decision record per LCU the encode pass ran the decision for (LCUs it shut SAO off for have none)
        assert (sao_recs["magic"] == 0x44414f53).all()
        extra["sao"] = sao_recs[np.lexsort((sao_recs["origin_x"], sao_recs["origin_y"], sao_recs["picture_number"]))]
    if costs and not inter:   # intra fixtures of the PM-core presets carry the pictures' rate tables too
        extra.update(cost_pictures=np.array(sorted(costs), np.uint64), cost=np.stack([costs[k] for k in sorted(costs)]))
    nl = S.lcu_count(w, h)
    order = np.lexsort((recs["lcu_index"], recs["picture_number"]))
    recs = recs[order]
    if inter:   # keep the P / B pictures only (the I picture is what the other fixtures hold) and what they read; the "sao_" sequences stay
        # whole: their pictures are each other's reference pictures (encode -> deblock -> SAO -> pad -> reference of the next picture)
        if not sao:
            recs = recs[recs["work"]["slice_type"] != 2]
        pocs = sorted(set(int(v) for v in recs["ref_poc"].reshape(-1) if v != 0xFFFFFFFFFFFFFFFF))
        assert pocs and all(k in refs for k in pocs), (pocs, sorted(refs))
        h0 = refs[pocs[0]][0]
        extra.update(ref_poc=recs["ref_poc"], ref_pocs=np.array(pocs, np.uint64),
                     ref_geom=np.array([int(h0[k]) for k in ("strideY", "strideC", "originX", "originY", "width", "height")], np.uint32),
                     ref_y=np.stack([refs[k][1][0] for k in pocs]), ref_cb=np.stack([refs[k][1][1] for k in pocs]),
                     ref_cr=np.stack([refs[k][1][2] for k in pocs]),
                     cost_pictures=np.array(sorted(costs), np.uint64), cost=np.stack([costs[k] for k in sorted(costs)]))
        assert all(int(k) in costs for k in set(recs["picture_number"][recs["work"]["slice_type"] != 2].tolist()))
    else:
        assert len(recs) == nl * n, "every LCU of an all-intra clip must be recorded (%d of %d)" % (len(recs), nl * n)
    path = os.path.join(S.GOLDEN_DIR, "encodepass_%s.npz" % name)
    np.savez_compressed(path, clip=np.array([kind, str(w), str(h), str(n), str(seed)]), enc_args=np.array(args),
                        picture_number=recs["picture_number"], lcu_index=recs["lcu_index"], dlf_off=recs["dlf_off"], work=recs["work"],
                        result=recs["result"], **extra)
    sizes, cnt = np.unique(np.concatenate([r["work"]["cu"]["size"][:r["work"]["num_cus"]] for r in recs]), return_counts=True)
    cbf = np.concatenate([r["result"]["cu"]["cbf"][:r["work"]["num_cus"]] for r in recs])
    odc = np.concatenate([r["result"]["cu"]["only_dc"][:r["work"]["num_cus"]] for r in recs])
    print("%-28s %d LCUs -> %s (%.0f KiB); unit sizes %s; cbf set %s of %d; DC-only %s" %
          (name, len(recs), os.path.basename(path), os.path.getsize(path) / 1024, dict(zip(sizes.tolist(), cnt.tolist())),
           cbf.sum(axis=0).tolist(), len(cbf), odc.sum(axis=0).tolist()))
    if inter:
        cus = np.concatenate([r["work"]["cu"][:r["work"]["num_cus"]] for r in recs])
        it = cus[cus["pred_mode"] == 1]
        print("    pictures %s; inter units %d of %d: AMVP %d merge %d skip %d; L0 %d L1 %d bi %d; reference pictures %s" %
              (sorted(set(recs["picture_number"].tolist())), len(it), len(cus), (it["inter_kind"] == 0).sum(), (it["inter_kind"] == 1).sum(),
               (it["inter_kind"] == 2).sum(), (it["inter_dir"] == 0).sum(), (it["inter_dir"] == 1).sum(), (it["inter_dir"] == 2).sum(), pocs))


if __name__ == "__main__":
    if not os.path.exists(S.REF_APP):
        sys.exit("oracle/_ref/SvtHevcEncApp_ref missing: run `make -C oracle ref` (needs /root/reference)")
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
