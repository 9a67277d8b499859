"""CPU checks of the drop-in boundary: libglowhip.so loads, exports every symbol include/glowhip.h
declares, and the ctypes signature table covers exactly those symbols.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

import pytorch_glow_amd as G
from pytorch_glow_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "glowhip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(glowhip_[a-z0-9_]+)\s*\(", src)))


def test_header_cites_reference_for_every_entry_point():
    src = open(HEADER).read()
    assert src.count("network/module.py:") + src.count("network/model.py:") >= 12


def test_library_builds_and_loads():
    assert os.path.exists(_lib.build()), "libglowhip.so missing after build"
    lib = G.lib()
    assert lib.glowhip_version() == 102
    assert lib.glowhip_last_error() is not None


def test_exports_match_header():
    syms = declared_symbols()
    assert len(syms) >= 20
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(handle, s), f"{s} declared in glowhip.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes signature table and header disagree"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (glowhip_[a-z0-9_]+)", out))
    assert exported == set(syms), exported ^ set(syms)


def test_layer_desc_layout_matches_header():
    # 8 int32 + 14 pointers, no padding surprises
    assert ctypes.sizeof(_lib.LayerDesc) == 8 * 4 + 14 * 8
    assert _lib.LayerDesc.an_bias.offset == 32 and _lib.LayerDesc.f4_logs.offset == 32 + 13 * 8


def test_host_only_entry_points_without_gpu():
    lib = G.lib()
    assert lib.glowhip_invconv_scratch_bytes(12) == 12 * 24 * 8
    assert lib.glowhip_plan_create(None, 0) is None
    assert b"empty layer list" in lib.glowhip_last_error()
    # a squeeze-only plan is pure host bookkeeping
    d = (_lib.LayerDesc * 2)()
    d[0].kind, d[0].C, d[0].H, d[0].W = _lib.LAYER_SQUEEZE, 3, 8, 8
    d[1].kind, d[1].C, d[1].H, d[1].W = _lib.LAYER_SQUEEZE, 12, 4, 4
    h = lib.glowhip_plan_create(d, 2)
    assert h
    out = (ctypes.c_int32 * 3)()
    assert lib.glowhip_plan_output_shape(h, 0, out) == 0 and list(out) == [48, 2, 2]
    assert lib.glowhip_plan_output_shape(h, 1, out) == 0 and list(out) == [3, 8, 8]
    assert lib.glowhip_plan_workspace_bytes(h, 4) >= 2 * 4 * 192 * 4
    lib.glowhip_plan_destroy(h)
    d[1].C = 13  # does not chain
    assert lib.glowhip_plan_create(d, 2) is None
    assert b"does not chain" in lib.glowhip_last_error()
