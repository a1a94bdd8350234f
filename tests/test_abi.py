"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/svt_hevc_amd.h
declares; Python-side struct views match the C layout."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

import svtlib as S


def _declared_symbols():
    text = open(os.path.join(S.ROOT, "include", "svt_hevc_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(svt_amd_\w+)\s*\(", text))
    names |= {"svt_amd_" + n for n in re.findall(r"^SVT_AMD_DECL_(?:TRANSFORM|INTRA|INTRA_ANG|MCP_UNI|MCP_RAW|MCP_CUNI|MCP_CRAW)\((\w+)", text, flags=re.M)}
    return sorted(names)


def test_header_symbols_exported():
    assert os.path.exists(S.PRODUCT_SO), "run `python __graft_entry__.py build` first"
    out = subprocess.check_output(["nm", "-D", "--defined-only", S.PRODUCT_SO], text=True)
    exported = set(line.split()[-1] for line in out.splitlines() if " T " in line)
    declared = _declared_symbols()
    assert len(declared) >= 190
    missing = [s for s in declared if s not in exported]
    assert not missing, "declared but not exported: %s" % missing
    stray = [s for s in exported if not s.startswith("svt_amd_")]
    assert not stray, "non-API symbols exported: %s" % stray


def test_library_loads_without_gpu():
    lib = C.CDLL(S.PRODUCT_SO)
    lib.svt_amd_version.restype = C.c_char_p
    assert b"gfx950" in lib.svt_amd_version()


def test_struct_layouts():
    assert C.sizeof(S.MeParams) == S.ME_PARAMS_DTYPE.itemsize == 108
    assert C.sizeof(S.MeCuResult) == S.ME_CU_DTYPE.itemsize == 24
    assert C.sizeof(S.MeLcuResult) == S.ME_LCU_DTYPE.itemsize == 3420
    assert S.DUMP_DTYPE.itemsize == 3592


def test_header_compiles_as_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "svt_hevc_amd.h"\n'
                   '_Static_assert(sizeof(SvtAmdMeParams) == 108, "params");\n'
                   '_Static_assert(sizeof(SvtAmdMeLcuResult) == 3420, "lcu result");\n'
                   'int main(void) { return 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(S.ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_no_gpu_means_loud_failure():
    """Without a device the batched layer must fail (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        return
    lib = S.load_product()
    ctx = C.c_void_p()
    rc = lib.svt_amd_context_create(0, 640, 384, 2, C.byref(ctx))
    assert rc != 0 and not ctx.value
    assert lib.svt_amd_last_error()
