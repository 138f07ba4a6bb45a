"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports exactly the
symbols include/pigeon_hip.h declares.  No compute calls (there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pigeon_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(hip_lib):
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(hip_lib, s), f"{s} declared in pigeon_hip.h but not exported"


def test_ctypes_signatures_cover_header(hip_lib):
    from pigeon_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert hip_lib.pg_abi_version() == 1


def test_no_gpu_fails_loudly(hip_lib):
    import torch
    from pigeon_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.PigeonHipError):
        _lib.require_gpu()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under pigeon_amd/ may import it (or the reference)."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "pigeon_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "/root/reference" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
