"""-m gpu: end-to-end drop-in check.  The reference encoder with its MotionEstimateLcu
replaced by ONE svt_amd_me_picture() call per picture and its OpenLoopIntraSearchLcu by ONE
svt_amd_ois_picture() call per picture (integration/svt_hook_me.c, linked with --wrap) must emit a .265 that is byte-identical to the unmodified reference's
(oracle/_ref/SvtHevcEncApp_ref) on the same YUV - the reference's own `asm_test`
criterion (Tests/SVT-HEVC_FunctionalTests.py:830-853) with a third leg."""
import hashlib
import os
import subprocess

import pytest

import svtlib as S

pytestmark = pytest.mark.gpu

HIP_APP = os.path.join(S.ROOT, "integration", "_build", "SvtHevcEncApp_hip")

CASES = [
    ("motion", 640, 384, 8, ["-encMode", "9", "-pred-struct", "0"]),
    ("motion", 640, 384, 9, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1"]),
    ("noise", 320, 256, 6, ["-encMode", "4"]),
    ("motion", 1920, 1080, 6, ["-encMode", "9", "-pred-struct", "0"]),
    ("flat", 1024, 768, 5, ["-encMode", "7", "-rc", "1", "-tbr", "2000000"]),
    # the same with one logical processor
    ("flat", 1024, 768, 5, ["-encMode", "7", "-rc", "1", "-tbr", "2000000", "-lp", "1"]),
    # all-intra 1080p encMode 10 (BASELINE configs[0]): OIS with 8x8 CUs on every picture, no ME at all
    ("motion", 1920, 1080, 4, ["-encMode", "10", "-intra-period", "0"]),
    # encMode 4 flat low-delay P: 35-mode OIS on P pictures, SSD sub-pel ME
    ("motion", 416, 240, 5, ["-encMode", "4", "-pred-struct", "0", "-hierarchical-levels", "0"]),
    # HME switched off, user-defined search area
    ("motion", 640, 384, 6, ["-encMode", "6", "-use-default-me-hme", "0", "-hme", "0", "-search-w", "24", "-search-h", "16"]),
    # 2 x 2 tiles (unrestricted motion vectors: the search may cross tile borders, the default)
    ("motion", 832, 480, 6, ["-encMode", "5", "-tile_col_cnt", "2", "-tile_row_cnt", "2"]),
    # 10-bit input: the front half works on the 8-bit MSB planes (BASELINE configs[3]/[4] class)
    ("motion10", 640, 384, 5, ["-encMode", "7", "-bit-depth", "10"]),
    # BASELINE configs[2]: 4K, encMode 7, random access with 2 hierarchical levels, SAO, 60 fps
    ("motion", 3840, 2160, 5, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1", "-fps", "60"]),
    # BASELINE configs[3]: the same with 10-bit input in the COMPRESSED format (8-bit planes + 2-bit planes packed four to a
    # byte, -compressed-ten-bit-format 1: EbEncHandle.c:3329-3601 copies them as they are, EncodePassPackLcu unpacks per LCU)
    ("motion10c", 3840, 2160, 5, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1", "-fps", "60",
                                  "-bit-depth", "10", "-compressed-ten-bit-format", "1"]),
    ("motion10c", 640, 384, 6, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1",
                                "-bit-depth", "10", "-compressed-ten-bit-format", "1"]),
    # BASELINE configs[4] class: 10-bit, encMode 4 (SSD sub-pel search on all 85 PUs, 8x8 refinement on reference pictures,
    # PM-core), 4 tile columns.  At 7680x4320 the REFERENCE itself needs more than ten minutes for three pictures on the GPU
    # box's 256 host threads (measured, profiles/r02_c), so the whole-encoder comparison runs the same switches at 1280x768;
    # the 8K size is covered through the C-ABI in tests/test_gpu_me.py::test_me_8k_m4_rows_match_oracle
    ("motion10", 1280, 768, 5, ["-encMode", "4", "-bit-depth", "10", "-tile_col_cnt", "4", "-fps", "60"]),
]


def _encode(app, yuv, w, h, n, args, out):
    r = subprocess.run([app, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-b", out] +
                       ([] if "-asm" in args else ["-asm", "1"]) + ([] if "-q" in args else ["-q", "32"]) + args, capture_output=True, text=True,
                       timeout=120 if w * h * n <= 1920 * 1080 * 8 else 400,   # seconds: a wedged encoder must not eat the GPU session (the largest cases take < 60 s)
                       # the product's default is the CLOSED LOOP on the device (svt_hook_cfg: SVT_HOOK_MD=pb, pool 16, 12 + 4 lanes).  The cases of this file that
                       # prove ONE binding at a time (front half, per-candidate full loops, reconstruction, prediction, SAO ...) switch it off unless they ask for it
                       # themselves; the closed-loop cases below run with the defaults - the configuration bench.py measures
                       env=dict({"SVT_HOOK_MD": "off"}, **dict(os.environ, SVT_HOOK_VERBOSE="1")))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return hashlib.md5(open(out, "rb").read()).hexdigest(), r.stderr


@pytest.mark.parametrize("kind,w,h,n,args", CASES)
def test_bitstream_identical_with_gpu_me(tmp_path, kind, w, h, n, args):
    assert os.path.exists(HIP_APP) and os.path.exists(S.REF_APP), \
        "integration/_build and oracle/_ref must be prebuilt (python __graft_entry__.py build, needs /root/reference)"
    yuv = str(tmp_path / "clip.yuv")
    if kind.endswith("10c"):
        S.write_clip10_compressed(yuv, kind[:-3], w, h, n, 7)
    elif kind.endswith("10"):
        S.write_clip10(yuv, kind[:-2], w, h, n, 7)
    else:
        S.write_clip(yuv, kind, w, h, n, 7)
    ref_md5, _ = _encode(S.REF_APP, yuv, w, h, n, args, str(tmp_path / "ref.265"))
    hip_md5, log = _encode(HIP_APP, yuv, w, h, n, args, str(tmp_path / "hip.265"))
    if "-rc" in args:
        # Rate control is the one configuration whose QPs depend on WHEN something arrives: the rate-control kernel takes picture-manager tasks and packetization feedback
        # in arrival order (Codec/EbRateControlProcess.c: RC_PICTURE_MANAGER_RESULT / RC_PACKETIZATION_FEEDBACK_RESULT) and a later picture's QP uses the sizes fed back so
        # far - the bindings change WHEN (threads wait for the device), so the bitstream of a rate-controlled encode is not a function of its inputs alone, not even with
        # -lp 1 (one logical processor, four dozen threads; this leg disagreed in one of three runs on a 256-thread box, profiles/r05_ar_gpu_tests.log).  No retry and no
        # excuse by repetition: the case gets a proof that does not depend on time.  SVT_HOOK_FRONT_VERIFY=1 - inside the same rate-controlled encode every
        # MotionEstimateLcu / OpenLoopIntraSearchLcu call is answered by the device AND by the reference code on the same inputs (rcMEdistortion included), the two
        # answers are compared: all of them, none may differ.  The md5 comparison stays for the runs whose timing agrees; a difference is printed, not hidden.
        import re
        rep = str(tmp_path / "verify_report.txt")
        os.environ["SVT_HOOK_FRONT_VERIFY"], os.environ["SVT_HOOK_REPORT"] = "1", rep
        try:
            _encode(HIP_APP, yuv, w, h, n, args, str(tmp_path / "verify.265"))
        finally:
            del os.environ["SVT_HOOK_FRONT_VERIFY"], os.environ["SVT_HOOK_REPORT"]
        m = re.search(r"front-half verification: (\d+) MotionEstimateLcu answers compared with the reference code's, (\d+) differ; (\d+) OpenLoopIntraSearchLcu answers compared, (\d+) differ",
                      open(rep).read())
        assert m, open(rep).read()[-1500:]
        me_n, me_bad, ois_n, ois_bad = (int(v) for v in m.groups())
        nl = S.lcu_count(w, h)
        assert (me_n, me_bad, ois_n, ois_bad) == ((n - 1) * nl, 0, n * nl, 0), m.group(0)
        if hip_md5 != ref_md5:
            # the md5 gate stays HARD unless the unmodified reference disagrees with ITSELF on this box (a reference-vs-reference control): five more reference
            # encodes of the same clip; the hooked bitstream must be one the reference produced, or the reference must have produced more than one - and then the case is
            # reported as an expected failure with both sets, never as a pass
            ref_set = {ref_md5} | {_encode(S.REF_APP, yuv, w, h, n, args, str(tmp_path / ("ref%d.265" % k)))[0] for k in range(5)}
            print("RC_TIMING: reference bitstreams %s, hooked %s (all %d + %d front-half answers equal to the reference code's)" % (sorted(ref_set), hip_md5, me_n, ois_n))
            if hip_md5 not in ref_set:
                assert len(ref_set) > 1, "bitstream differs from a reference that is deterministic here (%s vs %s)" % (hip_md5, ref_md5)
                pytest.xfail("rate-controlled encode: the unmodified reference produced %d different bitstreams in 6 runs on this box; the hooked one is a further one" % len(ref_set))
            hip_md5 = ref_md5 = next(iter(ref_set & {hip_md5}))
    # every MotionEstimateLcu call is redirected at link time (--wrap); the hook announces itself
    assert "svt_hook_me: motion estimation on svt-hevc_amd" in log, "hook inactive:\n" + log[-1000:]
    # one device OIS call per picture; ME for every non-intra picture
    assert log.count("svt_hook_me: OIS picture") == n, log[-1000:]
    n_intra = n if "-intra-period" in args else 1
    assert log.count("svt_hook_me: ME picture") == n - n_intra, log[-1000:]
    assert hip_md5 == ref_md5, "bitstream differs from the reference"
    assert os.path.getsize(str(tmp_path / "hip.265")) > 100


FULLLOOP_CASES = [
    ("motion", 416, 240, 4, ["-encMode", "9", "-pred-struct", "0"]),
    ("motion", 416, 240, 3, ["-encMode", "10", "-intra-period", "0"]),          # encMode 10 needs the 1080p class:
    ("motion", 416, 240, 5, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2"]),
    ("motion", 416, 240, 3, ["-encMode", "5", "-pred-struct", "0"]),            # 8x8 CUs: 4x4 chroma transform units
    # encMode 4: the P pictures' luma loop runs the PM-core quantiser.  The reference's AVX2 / SSE2 helpers of that path differ
    # from its C code ("There is Mismatch between ASM vs C !", EbTransforms.c:2848), so this one compares C_DEFAULT against C_DEFAULT
    ("motion", 416, 240, 4, ["-encMode", "4", "-pred-struct", "0", "-asm", "0"]),
    # encMode 3 random access: EVERY picture prices coefficients with the CABAC-context-updating estimator (coeffCabacUpdate:
    # full-depth pictures with chroma in the loop, EbEncDecProcess.c:2115-2123) on top of PM-core, luma and chroma
    ("motion", 416, 240, 5, ["-encMode", "3", "-pred-struct", "2", "-hierarchical-levels", "2", "-asm", "0"]),
]


@pytest.mark.parametrize("kind,w,h,n,args", FULLLOOP_CASES)
def test_bitstream_identical_with_gpu_full_loop(tmp_path, kind, w, h, n, args):
    """Same check with the mode decision's luma full loop (ProductFullLoop) and chroma full loop (FullLoop_R +
    CuFullDistortionFastTuMode_R) also answered by the device, one fused kernel call per candidate
    (SVT_HOOK_FULLLOOP=1): transform, quantisation, distortion, rate and cbf decision of every candidate CU come from
    svt_amd_full_loop_luma / svt_amd_full_loop_chroma."""
    if "-intra-period" in args:
        w, h = 1920, 1080
        n = 1
    yuv = str(tmp_path / "clip.yuv")
    S.write_clip(yuv, kind, w, h, n, 7)
    ref_md5, _ = _encode(S.REF_APP, yuv, w, h, n, args, str(tmp_path / "ref.265"))
    os.environ["SVT_HOOK_FULLLOOP"] = "1"
    os.environ["SVT_HOOK_REPORT"] = str(tmp_path / "report.txt")
    try:
        hip_md5, log = _encode(HIP_APP, yuv, w, h, n, args, str(tmp_path / "hip.265"))
    finally:
        del os.environ["SVT_HOOK_FULLLOOP"]
        del os.environ["SVT_HOOK_REPORT"]
    assert "svt_hook_me: luma full loop (ProductFullLoop) on the GPU" in log, log[-1000:]
    if "-intra-period" not in args:  # intra pictures of these presets leave chroma to the encode pass
        assert "svt_hook_me: chroma full loop (FullLoop_R + CuFullDistortionFastTuMode_R) on the GPU" in log, log[-1000:]
    assert hip_md5 == ref_md5, "bitstream differs from the reference"
    # no full-loop call is left to the reference code any more: the I picture of every sub-4K clip runs the
    # CABAC-context-updating estimator, which the device now does too
    import re
    rep = open(str(tmp_path / "report.txt")).read()
    m = re.search(r"full loop luma (\d+) \(left to the reference code (\d+)\) chroma (\d+) \((\d+)\)", rep)
    assert m, rep
    assert int(m.group(1)) > 0 and int(m.group(2)) == 0 and int(m.group(4)) == 0, rep
    c = re.search(r"CABAC-context-updating estimator on the GPU: full loop luma (\d+) chroma (\d+)", rep)
    assert c, rep
    if "3" == args[args.index("-encMode") + 1]:   # full-depth pictures with chroma in the loop: every picture of encMode 3
        assert int(c.group(1)) > 0 and int(c.group(2)) > 0, rep


RECON_CASES = [
    ("motion", 416, 240, 4, ["-encMode", "9", "-pred-struct", "0"]),
    ("noise", 192, 128, 2, ["-encMode", "1", "-intra-period", "0", "-q", "25"]),   # 4x4 luma units: inverse DST
    ("motion10", 416, 240, 3, ["-encMode", "7", "-bit-depth", "10"]),              # EncodeGenerateRecon16bit
]


@pytest.mark.parametrize("kind,w,h,n,args", RECON_CASES)
def test_bitstream_and_recon_identical_with_gpu_reconstruction(tmp_path, kind, w, h, n, args):
    """The final encode pass's transform-unit reconstruction (EncodeGenerateRecon / 16bit, reached through the global
    table EncodeGenerateReconFunctionPtr) answered by svt_amd_recon_tu (SVT_HOOK_RECON=1): the reference pictures of
    every later picture then come from the device, so both the bitstream and the reconstruction output must match."""
    yuv = str(tmp_path / "clip.yuv")
    if kind.endswith("10"):
        S.write_clip10(yuv, kind[:-2], w, h, n, 7)
    else:
        S.write_clip(yuv, kind, w, h, n, 7)
    ref_md5, _ = _encode(S.REF_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "ref.yuv")], str(tmp_path / "ref.265"))
    os.environ["SVT_HOOK_RECON"] = "1"
    try:
        hip_md5, log = _encode(HIP_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "hip.yuv")], str(tmp_path / "hip.265"))
    finally:
        del os.environ["SVT_HOOK_RECON"]
    assert "svt_hook_me: transform-unit reconstruction (EncodeGenerateRecon) on the GPU" in log, log[-1000:]
    assert hip_md5 == ref_md5, "bitstream differs from the reference"
    a, b = open(str(tmp_path / "ref.yuv"), "rb").read(), open(str(tmp_path / "hip.yuv"), "rb").read()
    assert len(a) > 1000 and a == b, "reconstruction output differs from the reference"


INTRA_CASES = [
    ("motion", 416, 240, 3, ["-encMode", "9", "-intra-period", "0"]),
    ("noise", 320, 256, 4, ["-encMode", "6", "-pred-struct", "0", "-q", "25", "-constrd-intra", "1"]),   # intra CUs among inter ones
    ("motion10", 416, 240, 2, ["-encMode", "7", "-intra-period", "0", "-bit-depth", "10"]),
    # B pictures above the base layer decide with the open-loop twin (IntraPredictionOl: source neighbours)
    ("noise", 200, 136, 9, ["-encMode", "4", "-pred-struct", "2", "-hierarchical-levels", "2", "-q", "26", "-asm", "0"]),
    # encMode 1: chroma in the mode decision's full loop (IntraPredictionCl asked for the chroma pair as well), 2 x 2 tiles
    ("noise", 640, 384, 2, ["-encMode", "1", "-pred-struct", "0", "-q", "28", "-tile_row_cnt", "2", "-tile_col_cnt", "2"]),
]


@pytest.mark.parametrize("kind,w,h,n,args", INTRA_CASES)
def test_bitstream_and_recon_identical_with_gpu_intra_prediction(tmp_path, kind, w, h, n, args):
    """The encode pass's intra prediction of every 8..32 prediction unit (reference-sample generation with availability,
    substitution and smoothing + the prediction itself) and the mode decision's closed-loop intra prediction of every candidate
    (IntraPredictionCl) answered by svt_amd_intra_pu (SVT_HOOK_INTRA=1)."""
    yuv = str(tmp_path / "clip.yuv")
    if kind.endswith("10"):
        S.write_clip10(yuv, kind[:-2], w, h, n, 7)
    else:
        S.write_clip(yuv, kind, w, h, n, 7)
    ref_md5, _ = _encode(S.REF_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "ref.yuv")], str(tmp_path / "ref.265"))
    os.environ["SVT_HOOK_INTRA"] = "1"
    try:
        hip_md5, log = _encode(HIP_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "hip.yuv")], str(tmp_path / "hip.265"))
    finally:
        del os.environ["SVT_HOOK_INTRA"]
    assert "svt_hook_me: encode-pass intra prediction (reference samples + prediction) on the GPU" in log, log[-1000:]
    assert "svt_hook_me: mode-decision intra prediction (IntraPredictionCl) on the GPU" in log, log[-1000:]
    if "-hierarchical-levels" in args:
        assert "svt_hook_me: open-loop mode-decision intra prediction (IntraPredictionOl) on the GPU" in log, log[-1000:]
    if args[:2] == ["-encMode", "1"]:   # encMode <= 2: intra 4x4 coding units
        assert "svt_hook_me: encode-pass intra 4x4 prediction on the GPU" in log, log[-1000:]
        assert "svt_hook_me: mode-decision intra 4x4 search prediction (Intra4x4IntraPredictionCl) on the GPU" in log, log[-1000:]
    assert hip_md5 == ref_md5, "bitstream differs from the reference"
    a, b = open(str(tmp_path / "ref.yuv"), "rb").read(), open(str(tmp_path / "hip.yuv"), "rb").read()
    assert len(a) > 1000 and a == b, "reconstruction output differs from the reference"


INTER_CASES = [
    ("motion", 416, 240, 5, ["-encMode", "9", "-pred-struct", "0"]),
    ("motion", 320, 192, 9, ["-encMode", "6", "-pred-struct", "2", "-hierarchical-levels", "2"]),   # bi-prediction, two lists
    ("motion10", 320, 192, 9, ["-encMode", "6", "-pred-struct", "2", "-hierarchical-levels", "2", "-bit-depth", "10"]),  # 16-bit driver
]


@pytest.mark.parametrize("kind,w,h,n,args", INTER_CASES)
def test_bitstream_and_recon_identical_with_gpu_inter_prediction(tmp_path, kind, w, h, n, args):
    """The encode pass's inter prediction of every prediction unit (EncodePassInterPrediction) and the mode decision's
    inter prediction of every candidate (Inter2Nx2NPuPredictionHevc) answered by svt_amd_inter_pu_batch from reference
    pictures resident on the device (SVT_HOOK_INTER=1)."""
    yuv = str(tmp_path / "clip.yuv")
    if kind.endswith("10"):
        S.write_clip10(yuv, kind[:-2], w, h, n, 7)
    else:
        S.write_clip(yuv, kind, w, h, n, 7)
    ref_md5, _ = _encode(S.REF_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "ref.yuv")], str(tmp_path / "ref.265"))
    os.environ["SVT_HOOK_INTER"] = "1"
    os.environ["SVT_HOOK_REPORT"] = str(tmp_path / "report.txt")
    try:
        hip_md5, log = _encode(HIP_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "hip.yuv")], str(tmp_path / "hip.265"))
    finally:
        del os.environ["SVT_HOOK_INTER"]
        del os.environ["SVT_HOOK_REPORT"]
    if kind.endswith("10"):
        assert "svt_hook_me: encode-pass inter prediction (EncodePassInterPrediction16bit) on the GPU" in log, log[-1000:]
    else:
        assert "svt_hook_me: encode-pass inter prediction (EncodePassInterPrediction) on the GPU" in log, log[-1000:]
    # the mode decision's inter prediction too - for a 10-bit encode from the 8 MSBs of the 16-bit reference pictures
    # (UnPackReferenceBlock path, svt_amd_inter_pu_batch_msb): no call is left to the reference code
    assert "svt_hook_me: mode-decision inter prediction (Inter2Nx2NPuPredictionHevc) on the GPU" in log, log[-1000:]
    import re
    rep = open(str(tmp_path / "report.txt")).read()
    m = re.search(r"Inter2Nx2NPuPredictionHevc\s+(\d+) calls left to the reference code", rep)
    assert m and int(m.group(1)) == 0, rep
    assert hip_md5 == ref_md5, "bitstream differs from the reference"
    a, b = open(str(tmp_path / "ref.yuv"), "rb").read(), open(str(tmp_path / "hip.yuv"), "rb").read()
    assert len(a) > 1000 and a == b, "reconstruction output differs from the reference"


SAO_CASES = [
    ("motion", 416, 240, 5, ["-encMode", "9", "-pred-struct", "0", "-hierarchical-levels", "0", "-sao", "1"]),
    ("motion", 416, 240, 9, ["-encMode", "5", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1", "-q", "30"]),
    ("motion10", 416, 240, 4, ["-encMode", "9", "-pred-struct", "0", "-hierarchical-levels", "0", "-sao", "1", "-bit-depth", "10", "-q", "20"]),
    ("motion", 640, 384, 3, ["-encMode", "3", "-pred-struct", "0", "-hierarchical-levels", "0", "-tile_row_cnt", "2", "-tile_col_cnt", "2"]),
]


@pytest.mark.parametrize("kind,w,h,n,args", SAO_CASES)
def test_bitstream_and_recon_identical_with_gpu_sao_decision(tmp_path, kind, w, h, n, args):
    """Every LCU's SAO statistics and parameter decision (SaoGenerationDecision / SaoGenerationDecision16bit) answered by the
    device (SVT_HOOK_SAO=1): gather entry points + svt_amd_sao_decide_lcu."""
    yuv = str(tmp_path / "clip.yuv")
    if kind.endswith("10"):
        S.write_clip10(yuv, kind[:-2], w, h, n, 7)
    else:
        S.write_clip(yuv, kind, w, h, n, 7)
    ref_md5, _ = _encode(S.REF_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "ref.yuv")], str(tmp_path / "ref.265"))
    os.environ["SVT_HOOK_SAO"] = "1"
    try:
        hip_md5, log = _encode(HIP_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "hip.yuv")], str(tmp_path / "hip.265"))
    finally:
        del os.environ["SVT_HOOK_SAO"]
    assert "svt_hook_me: SAO statistics + decision (SaoGenerationDecision" in log, log[-1000:]
    assert hip_md5 == ref_md5, "bitstream differs from the reference"
    a, b = open(str(tmp_path / "ref.yuv"), "rb").read(), open(str(tmp_path / "hip.yuv"), "rb").read()
    assert len(a) > 1000 and a == b, "reconstruction output differs from the reference"


def test_bitstream_identical_with_pm_core_on_the_device(tmp_path):
    """encMode 4 (BASELINE configs[4]'s preset): the PM-core variants of the luma full loop (P pictures) and of the encode-pass
    quantiser (every picture) on the device, C_DEFAULT against C_DEFAULT (the reference's AVX2 helpers of this path differ
    from its C code, EbTransforms.c:2848)."""
    kind, w, h, n, args = "motion", 416, 240, 4, ["-encMode", "4", "-pred-struct", "0", "-asm", "0"]
    yuv = str(tmp_path / "clip.yuv")
    S.write_clip(yuv, kind, w, h, n, 7)
    ref_md5, _ = _encode(S.REF_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "ref.yuv")], str(tmp_path / "ref.265"))
    flags = ("SVT_HOOK_FULLLOOP", "SVT_HOOK_QUANT")
    for f in flags:
        os.environ[f] = "1"
    try:
        hip_md5, log = _encode(HIP_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "hip.yuv")], str(tmp_path / "hip.265"))
    finally:
        for f in flags:
            del os.environ[f]
    assert "luma full loop (ProductFullLoop) on the GPU" in log and "encode-pass PM-core quantiser" in log, log[-1500:]
    assert hip_md5 == ref_md5, "bitstream differs from the reference"
    assert open(str(tmp_path / "ref.yuv"), "rb").read() == open(str(tmp_path / "hip.yuv"), "rb").read()


@pytest.mark.parametrize("kind,w,h,n,args", [("motion", 416, 240, 5, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2"]),
                                             ("motion10", 416, 240, 3, ["-encMode", "9", "-pred-struct", "0", "-bit-depth", "10"])])
def test_bitstream_identical_with_every_binding_enabled(tmp_path, kind, w, h, n, args):
    """All device bindings at once: ME, OIS, both MD full loops, encode-pass intra and inter prediction, quantiser,
    transform-unit reconstruction and the SAO statistics + decision."""
    yuv = str(tmp_path / "clip.yuv")
    if kind.endswith("10"):
        S.write_clip10(yuv, kind[:-2], w, h, n, 7)
    else:
        S.write_clip(yuv, kind, w, h, n, 7)
    ref_md5, _ = _encode(S.REF_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "ref.yuv")], str(tmp_path / "ref.265"))
    flags = ("SVT_HOOK_FULLLOOP", "SVT_HOOK_RECON", "SVT_HOOK_INTRA", "SVT_HOOK_INTER", "SVT_HOOK_QUANT", "SVT_HOOK_SAO")
    for f in flags:
        os.environ[f] = "1"
    try:
        hip_md5, log = _encode(HIP_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "hip.yuv")], str(tmp_path / "hip.265"))
    finally:
        for f in flags:
            del os.environ[f]
    for msg in ("motion estimation on svt-hevc_amd", "luma full loop (ProductFullLoop) on the GPU", "chroma full loop",
                "transform-unit reconstruction", "encode-pass intra prediction", "encode-pass inter prediction",
                "encode-pass quantiser", "SAO statistics + decision"):
        if kind.endswith("10") and "inter prediction" in msg:
            msg = "encode-pass inter prediction (EncodePassInterPrediction16bit)"
        assert msg in log, (msg, log[-1500:])
    assert hip_md5 == ref_md5, "bitstream differs from the reference"
    assert open(str(tmp_path / "ref.yuv"), "rb").read() == open(str(tmp_path / "hip.yuv"), "rb").read()


ENCODEPASS_CASES = [
    # all-intra, default loop filters (deblocking + SAO run in the reference code on the device's reconstruction); partial LCUs
    ("motion", 416, 240, 3, ["-encMode", "9", "-intra-period", "0"], "all"),
    ("noise", 200, 136, 2, ["-encMode", "6", "-intra-period", "0", "-q", "22"], "all"),
    # BASELINE configs[0]: all-intra 1080p encMode 10
    ("motion", 1920, 1080, 2, ["-encMode", "10", "-intra-period", "0"], "all"),
    # random access: I, P and B pictures on the device - inter units (uni / bi-prediction from device copies of the reference pictures,
    # merge / skip / AMVP) and the intra units between them
    ("motion", 640, 384, 9, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1"], "inter"),
    # low-delay P with constrained intra prediction: inter neighbours are unavailable to intra units; at encMode 8 the top-layer
    # pictures re-decide merge / skip with chroma inside EncodePass (CHROMA_MODE_BEST): the binding runs that host step before the call
    ("noise", 320, 256, 4, ["-encMode", "8", "-pred-struct", "0", "-constrd-intra", "1", "-q", "40"], "full"),
    # encMode 9 random access, 4 temporal layers: non-reference B pictures with CHROMA_MODE_BEST, skip-cost bias (:3865-3878)
    ("motion", 416, 240, 9, ["-encMode", "9", "-pred-struct", "2", "-hierarchical-levels", "3"], "inter"),
    # low-delay P, flat prediction structure, 64x64 units at a high QP, AMVP units on noise
    ("motion", 416, 240, 6, ["-encMode", "6", "-pred-struct", "0", "-hierarchical-levels", "0", "-q", "40"], "inter"),
    ("noise", 200, 136, 5, ["-encMode", "5", "-pred-struct", "1", "-hierarchical-levels", "0", "-q", "46"], "inter"),
    # 2 x 2 tiles: tile edges cut the intra neighbourhood, four wavefronts share the picture
    ("motion", 832, 480, 3, ["-encMode", "7", "-intra-period", "0", "-tile_col_cnt", "2", "-tile_row_cnt", "2"], "all"),
    # 2 x 2 tiles with P / B pictures: four wavefronts, inter units next to tile edges
    ("motion", 832, 480, 9, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-tile_col_cnt", "2", "-tile_row_cnt", "2"], "inter"),
    # encMode 4 / 3: the encode pass quantises with PM-core (luma levels re-decided per 4x4 block), on the device too
    ("motion", 416, 240, 2, ["-encMode", "4", "-intra-period", "0"], "all"),
    ("motion", 416, 240, 6, ["-encMode", "3", "-pred-struct", "2", "-hierarchical-levels", "2"], "inter"),
    # encMode 2: LCUs with intra 4x4 units stay on the host between device-encoded ones
    ("noise", 200, 136, 3, ["-encMode", "2", "-pred-struct", "0", "-hierarchical-levels", "0", "-q", "26"], "mixed"),
    # 10-bit encodes: EncodePass with is16bit through the 16-bit contract (all-intra, and random access with host-encoded LCUs)
    ("motion10", 416, 240, 3, ["-encMode", "9", "-intra-period", "0", "-bit-depth", "10"], "all"),
    ("noise10", 320, 256, 4, ["-encMode", "8", "-pred-struct", "0", "-constrd-intra", "1", "-q", "40", "-bit-depth", "10"], "full"),
    ("motion10c", 640, 384, 5, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1", "-bit-depth", "10",
                                "-compressed-ten-bit-format", "1"], "inter"),
]


@pytest.mark.parametrize("kind,w,h,n,args,expect", ENCODEPASS_CASES)
def test_bitstream_and_recon_identical_with_device_resident_encode_pass(tmp_path, kind, w, h, n, args, expect):
    """SVT_HOOK_ENCODEPASS=1: EncodePass of every LCU inside the device call's coverage is ONE svt_amd_encode_lcus() call (prediction,
    transform, quantiser, reconstruction of all its units against the picture's reconstruction in HBM); the reference's own
    EncodePass then only keeps its books from the output contract (integration/svt_hook_encdec.c).  Bitstream and reconstruction
    must be byte-identical to the unmodified reference's."""
    import re
    yuv = str(tmp_path / "clip.yuv")
    if kind.endswith("10c"):
        S.write_clip10_compressed(yuv, kind[:-3], w, h, n, 7)
    elif kind.endswith("10"):
        S.write_clip10(yuv, kind[:-2], w, h, n, 7)
    else:
        S.write_clip(yuv, kind, w, h, n, 7)
    ref_md5, _ = _encode(S.REF_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "ref.yuv")], str(tmp_path / "ref.265"))
    os.environ["SVT_HOOK_ENCODEPASS"] = "1"
    os.environ["SVT_HOOK_REPORT"] = str(tmp_path / "report.txt")
    try:
        hip_md5, log = _encode(HIP_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "hip.yuv")], str(tmp_path / "hip.265"))
    finally:
        del os.environ["SVT_HOOK_ENCODEPASS"]
        del os.environ["SVT_HOOK_REPORT"]
    rep = open(str(tmp_path / "report.txt")).read()
    m = re.search(r"encode pass: (\d+) LCUs encoded on the GPU \(one call each; (\d+) of them with inter units, (\d+) inter units\); left to the "
                  r"reference code: (\d+) LCUs with units outside the device call, (\d+) under tools outside it, (\d+) in another sample format; "
                  r"(\d+) host LCU borders handed over in (\d+) calls", rep)
    assert m, rep
    gpu, inter_lcus, inter_units, units, tools, fmt, borders, puts = (int(x) for x in m.groups())
    nl = S.lcu_count(w, h) * n
    assert gpu + units + tools + fmt == nl, rep
    if expect == "all":
        assert gpu == nl, rep
    elif expect == "inter":    # every LCU of every picture, most of them with inter units
        assert gpu == nl and inter_lcus > nl // 3 and inter_units >= inter_lcus, rep
    elif expect == "full":     # every LCU, some with inter units (noise: most units are intra)
        assert gpu == nl and inter_lcus > 0, rep
    elif expect == "mixed":    # device-encoded LCUs (with inter units too) after host-encoded ones inside P pictures
        assert gpu > S.lcu_count(w, h) and inter_lcus > 0 and units > 0 and borders == units and puts <= units, rep
    else:
        assert gpu == 0 and tools == nl, rep
    assert hip_md5 == ref_md5, "bitstream differs from the reference\n" + rep
    assert open(str(tmp_path / "ref.yuv"), "rb").read() == open(str(tmp_path / "hip.yuv"), "rb").read()


VERIFY_CASES = [
    ("motion", 1920, 1080, 13, ["-encMode", "7"]),
    ("noise", 832, 480, 9, ["-encMode", "5", "-pred-struct", "1", "-q", "38"]),
    ("motion10", 1280, 720, 9, ["-encMode", "8", "-bit-depth", "10"]),
]


@pytest.mark.parametrize("kind,w,h,n,args", VERIFY_CASES)
def test_encode_pass_verification_mode_finds_no_difference(tmp_path, kind, w, h, n, args):
    """SVT_HOOK_ENCODEPASS_VERIFY=1: every LCU is encoded on the device AND by the reference's own EncodePass inside the running encoder;
    flags, coefficients and (loop filters off) the un-deblocked reconstruction of every unit are compared there
    (integration/svt_hook_encdec.c:verify_lcu) - thousands of LCUs of real P / B pictures per case"""
    import re
    yuv = str(tmp_path / "clip.yuv")
    if kind.endswith("10"):
        S.write_clip10(yuv, kind[:-2], w, h, n, 7)
    else:
        S.write_clip(yuv, kind, w, h, n, 7)
    os.environ["SVT_HOOK_ENCODEPASS"] = "1"
    os.environ["SVT_HOOK_ENCODEPASS_VERIFY"] = "1"
    os.environ["SVT_HOOK_REPORT"] = str(tmp_path / "report.txt")
    try:
        _, log = _encode(HIP_APP, yuv, w, h, n, args + ["-dlf", "1", "-sao", "0"], str(tmp_path / "hip.265"))
    finally:
        for k in ("SVT_HOOK_ENCODEPASS", "SVT_HOOK_ENCODEPASS_VERIFY", "SVT_HOOK_REPORT"):
            del os.environ[k]
    rep = open(str(tmp_path / "report.txt")).read()
    m = re.search(r"encode pass verification: (\d+) device-encoded LCUs compared with the reference's own EncodePass, (\d+) differ", rep)
    assert m, rep
    compared, differ = int(m.group(1)), int(m.group(2))
    assert compared == S.lcu_count(w, h) * n and differ == 0, rep + "\n".join(ln for ln in log.splitlines() if "VERIFY" in ln)[:3000]
    assert re.search(r"one call each; (\d+) of them with inter units", rep) and int(re.search(r"one call each; (\d+) of them", rep).group(1)) > compared // 4, rep


DEVICE_REF_CASES = [
    ("motion", 640, 384, 9, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1"]),
    ("motion", 416, 240, 8, ["-encMode", "6", "-pred-struct", "0", "-hierarchical-levels", "0", "-q", "36"]),
    ("motion10", 640, 384, 5, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1", "-bit-depth", "10"]),
    ("motion", 832, 480, 5, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-tile_col_cnt", "2", "-tile_row_cnt", "2"]),
    # encMode 9, 4 temporal layers: the reference pictures of layers 1 and 2 are "encoder / decoder mismatch" pictures (no deblocking, no
    # SAO on the encoder side): the device finishes them as the encoder does
    ("motion", 640, 384, 9, ["-encMode", "9", "-pred-struct", "2", "-hierarchical-levels", "3"]),
]


@pytest.mark.parametrize("kind,w,h,n,args", DEVICE_REF_CASES)
def test_reference_pictures_finished_on_the_device_inside_the_encoder(tmp_path, kind, w, h, n, args):
    """SVT_HOOK_ENCODEPASS=1 SVT_HOOK_ENCODEPASS_REFS=verify: when the last LCU of a reference picture has been encoded on the device, the
    device finishes the picture itself - deblocking, SAO (statistics in the encoder's order, decision, application), padding - and the
    result IS the reference picture later pictures predict from (device-to-device into the reference cache, never uploaded).  Each one is
    compared once with the reference picture the encoder finished on the host; the bitstream must stay byte-identical."""
    import re
    yuv = str(tmp_path / "clip.yuv")
    (S.write_clip10 if kind.endswith("10") else S.write_clip)(yuv, kind[:-2] if kind.endswith("10") else kind, w, h, n, 7)
    ref_md5, _ = _encode(S.REF_APP, yuv, w, h, n, args, str(tmp_path / "ref.265"))
    env = {"SVT_HOOK_ENCODEPASS": "1", "SVT_HOOK_ENCODEPASS_REFS": "verify", "SVT_HOOK_REPORT": str(tmp_path / "report.txt")}
    os.environ.update(env)
    try:
        hip_md5, log = _encode(HIP_APP, yuv, w, h, n, args, str(tmp_path / "hip.265"))
    finally:
        for k in env:
            del os.environ[k]
    rep = open(str(tmp_path / "report.txt")).read()
    m = re.search(r"(\d+) reference pictures finished on the device \(encode pass -> deblocking -> SAO -> padding\) and never uploaded; (\d+) compared "
                  r"with the encoder's own, (\d+) differ", rep)
    assert m, rep
    made, compared, differ = (int(v) for v in m.groups())
    assert made >= 2 and compared >= 1 and differ == 0, rep
    assert hip_md5 == ref_md5, "bitstream differs from the reference\n" + rep


MD_CASES = [
    # all-intra: EVERY picture's mode decision + encode pass is one device call
    ("motion", 416, 240, 3, ["-encMode", "9", "-intra-period", "0"], "all"),
    ("noise", 320, 256, 2, ["-encMode", "8", "-intra-period", "0", "-q", "26"], "all"),
    # BASELINE configs[0]: 1080p encMode 10 all-intra
    ("motion", 1920, 1080, 3, ["-encMode", "10", "-intra-period", "0"], "all"),
    # 2 x 2 tiles
    ("motion", 640, 384, 2, ["-encMode", "9", "-intra-period", "0", "-tile_col_cnt", "2", "-tile_row_cnt", "2"], "all"),
    # random access, encMode 9: the I picture, the non-reference B pictures (open-loop intra, luma-only candidates) and the reference B picture of temporal layer 1
    # (chroma level 4: CHROMA_MODE_FULL LCUs, chroma in both loops) on the device - here pictures 0, 1, 3, 5 and 2; the base-layer B picture (closed-loop intra,
    # branch-and-depth-pillar LCUs) stays with the reference code
    ("motion", 640, 384, 6, ["-encMode", "9", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1"], ("inter", 5, 4)),
    # encMode 8, three hierarchical levels, moving objects (AMVP, uni- and bi-prediction, merge / skip decisions with chroma): I + the 4 non-reference B pictures + the
    # reference B pictures of layers 1 and 2 whose LCUs all take the ModeDecisionLcu path
    ("objects", 416, 240, 9, ["-encMode", "8", "-pred-struct", "2", "-hierarchical-levels", "3", "-q", "28"], ("inter", 8, 7)),
    # low delay P
    ("objects", 320, 192, 6, ["-encMode", "8", "-pred-struct", "0", "-hierarchical-levels", "2", "-q", "30"], ("inter", 5, 4)),
    # noise: intra units inside B pictures
    ("noise", 320, 256, 5, ["-encMode", "9", "-pred-struct", "2", "-hierarchical-levels", "2", "-q", "24"], ("inter", 4, 3)),
    # encMode 6 (chroma in the mode decision, CABAC-context update): outside this revision, every picture left to the reference code
    ("motion", 416, 240, 2, ["-encMode", "6", "-intra-period", "0"], "none"),
    # 10-bit (BASELINE configs[3]'s bit depth and format at a small size): the mode decision on the 8 MSBs of source and reference pictures, the encode pass on the
    # 10-bit samples, one svt_amd_md_encode_picture[_inter]16 call per picture; unpacked and compressed ("packed") 2-bit planes
    ("motion10", 416, 240, 3, ["-encMode", "9", "-intra-period", "0", "-bit-depth", "10"], "all"),
    ("motion10c", 640, 384, 6, ["-encMode", "9", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1", "-bit-depth", "10", "-compressed-ten-bit-format", "1"],
     ("inter", 5, 4)),
    ("objects10", 416, 240, 9, ["-encMode", "8", "-pred-struct", "2", "-hierarchical-levels", "3", "-q", "28", "-bit-depth", "10"], ("inter", 8, 7)),
]


@pytest.mark.parametrize("kind,w,h,n,args,expect", MD_CASES)
def test_bitstream_and_recon_identical_with_device_resident_mode_decision(tmp_path, kind, w, h, n, args, expect):
    """SVT_HOOK_MD=1: ModeDecisionLcu AND EncodePass of every LCU of a covered picture are ONE svt_amd_md_encode_picture() call made by the
    picture's first ModeDecisionLcu call (candidate lists, fast loop, full loop, inter-depth decision, neighbour state and the encode pass all
    on the device, wavefront included); the reference's per-LCU calls then only copy decisions and keep books.  Bitstream and reconstruction
    must be byte-identical to the unmodified reference's."""
    import re
    yuv = str(tmp_path / "clip.yuv")
    if kind.endswith("10c"):
        S.write_clip10_compressed(yuv, kind[:-3], w, h, n, 7)
    elif kind.endswith("10"):
        S.write_clip10(yuv, kind[:-2], w, h, n, 7)
    else:
        S.write_clip(yuv, kind, w, h, n, 7)
    ref_md5, _ = _encode(S.REF_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "ref.yuv")], str(tmp_path / "ref.265"))
    env = {"SVT_HOOK_MD": "1", "SVT_HOOK_REPORT": str(tmp_path / "report.txt")}
    os.environ.update(env)
    try:
        hip_md5, log = _encode(HIP_APP, yuv, w, h, n, args + ["-o", str(tmp_path / "hip.yuv")], str(tmp_path / "hip.265"))
    finally:
        for k in env:
            del os.environ[k]
    rep = open(str(tmp_path / "report.txt")).read()
    m = re.search(r"mode decision: (\d+) pictures \((\d+) of them P / B; (\d+) LCUs\) decided AND encoded by ONE device call each .*?; (\d+) pictures outside the "
                  r"device call", rep)
    assert m, rep
    pics, inter, lcus, left = (int(v) for v in m.groups())
    nl = S.lcu_count(w, h)
    if expect == "all":
        assert pics == n and lcus == n * nl and left == 0, rep
    elif isinstance(expect, tuple):
        assert inter >= 1 and pics == inter + 1 and lcus == pics * nl, rep   # the I picture + the P / B pictures inside the device call
        print("MD_COUNTS", kind, w, h, n, pics, inter, left)   # (pytest -s: the exact counts of a run, for pinning)
        if expect[1] is not None:   # exactly these (read off a run with pytest -s, profiles/r05_c; the counts repeat): which pictures qualify follows from the LCU depth modes the reference derives for this clip - a fixed function of the clip
            assert pics == expect[1] and inter == expect[2], rep
    else:
        assert pics == 0 and left >= 1, rep
    assert hip_md5 == ref_md5, "bitstream differs from the reference\n" + rep
    assert open(str(tmp_path / "ref.yuv"), "rb").read() == open(str(tmp_path / "hip.yuv"), "rb").read()


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4"])
def test_baseline_config2_with_the_device_closed_loop_on_is_bitstream_identical(tmp_path, cfg):
    """BASELINE configs[2] itself (4K, encMode 7, random access, SAO on; tools/encoder_fps.py "cfg3") and configs[3] (the same in 10-bit, packed 2-bit planes; "cfg4"),
    17 pictures: motion estimation + open-loop intra search on the device and, with SVT_HOOK_MD=1, mode decision + encode pass of the I picture, of every non-reference
    B picture (temporal layer 2) and of every layer-1 reference B picture (CHROMA_MODE_FULL) as ONE device call each (10-bit: the mode decision on the 8 MSBs, the encode
    pass on the 10-bit samples) - the bitstream must be the unmodified reference's"""
    import re
    import sys
    sys.path.insert(0, os.path.join(S.ROOT, "tools"))
    import encoder_fps as E
    rp = str(tmp_path / "report.txt")
    r = E.measure(cfg, frames=17, hip_env={"SVT_HOOK_MD": "1", "SVT_HOOK_REPORT": rp}, tmpdir=str(tmp_path))
    rep = open(rp).read()
    m = re.search(r"mode decision: (\d+) pictures \((\d+) of them P / B; (\d+) LCUs\)", rep)
    assert m, rep
    pics, inter, lcus = (int(v) for v in m.groups())
    assert pics == 13 and inter == 12 and lcus == pics * S.lcu_count(3840, 2160), rep   # I + 8 layer-2 + 4 layer-1 pictures; the 4 base-layer B pictures: reference code
    assert r["bitstream_identical"], rep


def test_baseline_config1_with_the_device_closed_loop_on_is_bitstream_identical(tmp_path):
    """BASELINE configs[1] (1080p, encMode 9, low-delay P, 64 frames; tools/encoder_fps.py "cfg2") with SVT_HOOK_MD=1 and the pool / limiter settings the bench runs with:
    the bitstream must be the unmodified reference's, and the binding's report says which pictures took the device call - every picture is accounted for (decided on the
    device, or left to the reference code because its LCUs take the branch-and-depth-pillar path: the base layer of the low-delay structure), none silently."""
    import re
    import sys
    sys.path.insert(0, os.path.join(S.ROOT, "tools"))
    import encoder_fps as E
    rp = str(tmp_path / "report.txt")
    frames = 64
    r = E.measure("cfg2", frames=frames, hip_env={"SVT_HOOK_MD": "1", "SVT_HOOK_PCS_POOL": "8", "SVT_HOOK_REPORT": rp}, tmpdir=str(tmp_path))
    rep = open(rp).read()
    m = re.search(r"mode decision: (\d+) pictures \((\d+) of them P / B; (\d+) LCUs\).*?; (\d+) pictures outside", rep)
    assert m, rep
    pics, inter, lcus, left = (int(v) for v in m.groups())
    print("MD_COUNTS cfg2", frames, pics, inter, left)
    assert lcus == pics * S.lcu_count(1920, 1080), rep
    assert pics >= 1 and pics == inter + 1, rep                       # the I picture + the P pictures whose LCUs all take ModeDecisionLcu
    assert (pics, inter, left) == (57, 56, 7), rep                     # the I picture + 56 of the 63 P pictures; the 7 base-layer pictures (every 8th: closed-loop intra + BDP LCUs) are the reference code's
    assert r["bitstream_identical"], rep


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4"])
def test_baseline_config2_with_the_library_defaults_is_bitstream_identical(tmp_path, cfg):
    """What a host gets by LOADING the drop-in library and nothing else (no SVT_HOOK_* switch in the environment): the configuration bench.py measures - closed loop of the
    P / B pictures on the device (SVT_HOOK_MD=pb), 16 picture control sets in the EncDec pool, 12 EncDec + 4 front-half lanes - at BASELINE configs[2] / [3], 33 pictures
    at the bench's -lp 32: bitstream identical, 24 of 33 pictures (every layer-1 and layer-2 B picture) decided and encoded by the device call, the I picture and the
    8 base-layer B pictures by the reference's code."""
    import re
    import sys
    sys.path.insert(0, os.path.join(S.ROOT, "tools"))
    import encoder_fps as E
    rp = str(tmp_path / "report.txt")
    assert not [k for k in os.environ if k.startswith("SVT_HOOK_")], "this case runs the library's defaults"
    r = E.measure(cfg, frames=33, extra=["-lp", "32"], hip_env={"SVT_HOOK_REPORT": rp}, tmpdir=str(tmp_path))
    rep = open(rp).read()
    m = re.search(r"mode decision: (\d+) pictures \((\d+) of them P / B; (\d+) LCUs\).*?; (\d+) pictures outside", rep)
    assert m, rep
    pics, inter, lcus, left = (int(v) for v in m.groups())
    assert (pics, inter, left) == (24, 24, 8) and lcus == pics * S.lcu_count(3840, 2160), rep   # left: the 8 base-layer B pictures (the I picture is not offered to the device call at SVT_HOOK_MD=pb)
    assert r["bitstream_identical"], rep


def _encode_with_report(tmp_path, yuv, w, h, n, args, env):
    rep = str(tmp_path / "report.txt")
    r = subprocess.run([HIP_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-n", str(n), "-b", str(tmp_path / "hip.265"), "-asm", "1", "-q", "32"] + args,
                       capture_output=True, text=True, timeout=300, env=dict({"SVT_HOOK_MD": "off"}, **dict(os.environ, SVT_HOOK_REPORT=rep, **env)))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return hashlib.md5(open(str(tmp_path / "hip.265"), "rb").read()).hexdigest(), open(rep).read()


@pytest.mark.parametrize("kind,w,h,n,args", [
    ("objects", 640, 384, 9, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1", "-intra-period", "3"]),
    ("motion", 1920, 1080, 5, ["-encMode", "5", "-pred-struct", "2", "-hierarchical-levels", "3"]),
    ("objects", 416, 240, 6, ["-encMode", "9", "-intra-period", "0"]),
])
def test_bitstream_identical_with_the_ac_energy_of_i_pictures_from_the_device(tmp_path, kind, w, h, n, args):
    """SURVEY 8f-3 bound in the encoder (SVT_HOOK_SBO=1): every ComputeNxMSatdSadLCU call of CalculateAcEnergy (EbSourceBasedOperationsProcess.c:323-339) is answered
    from ONE svt_amd_picture_ac_energy launch per I picture on the luma the front half uploaded; the values steer DeriveDefaultSegments and the encode pass's
    contour tests, so a wrong one moves the stream."""
    import re
    yuv = str(tmp_path / "clip.yuv")
    S.write_clip(yuv, kind, w, h, n, 7)
    ref_md5, _ = _encode(S.REF_APP, yuv, w, h, n, args, str(tmp_path / "ref.265"))
    hip_md5, report = _encode_with_report(tmp_path, yuv, w, h, n, args, {"SVT_HOOK_SBO": "1"})
    m = re.search(r"AC energy \(CalculateAcEnergy\) of (\d+) pictures on the GPU, (\d+) ComputeNxMSatdSadLCU calls answered from it, (\d+) left", report)
    assert m, report[-1500:]
    pictures, answered, left = (int(x) for x in m.groups())
    print("SBO_COUNTS", kind, w, h, n, pictures, answered, left)
    assert pictures >= 1 and answered == pictures * 5 * (w // 64) * (h // 64) and left == 0
    assert hip_md5 == ref_md5, "bitstream differs from the reference"


@pytest.mark.parametrize("w,h,n,args", [
    (640, 384, 5, ["-encMode", "7", "-pred-struct", "2", "-hierarchical-levels", "2", "-sao", "1", "-bit-depth", "10"]),
    (1920, 1080, 3, ["-encMode", "9", "-pred-struct", "0", "-bit-depth", "10"]),
    (424, 240, 4, ["-encMode", "9", "-bit-depth", "10"]),                      # a width the row bands' 8-sample groups do not divide
])
def test_bitstream_identical_with_the_16_bit_input_unpacked_on_the_device(tmp_path, w, h, n, args):
    """SURVEY 8f-4 bound in the encoder (SVT_HOOK_UNPACK=1): the UnPack2D threads (EbPictureOperators.c:512) keep the reference's job protocol and split every row
    band of the application's 16-bit planes into the 8-bit + 2-bit planes on the device."""
    import re
    yuv = str(tmp_path / "clip.yuv")
    S.write_clip10(yuv, "motion", w, h, n, 7)
    ref_md5, _ = _encode(S.REF_APP, yuv, w, h, n, args, str(tmp_path / "ref.265"))
    hip_md5, report = _encode_with_report(tmp_path, yuv, w, h, n, args, {"SVT_HOOK_UNPACK": "1"})
    m = re.search(r"\(UnPack2D\) on the GPU: (\d+) jobs, (\d+) samples", report)
    assert m, report[-1500:]
    assert int(m.group(1)) >= 3 * n and 0.9 * n * w * h * 1.5 <= int(m.group(2)) <= n * w * h * 3 // 2
    assert hip_md5 == ref_md5, "bitstream differs from the reference"
