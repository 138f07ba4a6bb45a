"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports exactly the
symbols include/pigeon_hip.h declares.  No compute calls (there is no GPU here)."""
import os
import re
import sys

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
    assert hip_lib.pg_abi_version() == 5


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


def test_every_device_kernel_has_its_host_side(hip_lib):
    """A kernel template whose body (or a lambda in it) contains something that is not valid HOST code -- an address-space cast,
    a "v" asm constraint, a buffer-resource parameter of a template -- loses its host side WITHOUT a compiler diagnostic; the
    library then links (shared objects may have undefined symbols) and only fails at dlopen, on the GPU box (round 3, the first
    16x16x32 attention kernel).  Loading the library in the fixture already proves the product build; this also loads the tools
    build when it exists, and checks both with `nm -u` for undefined kernel stubs."""
    import ctypes
    import subprocess
    libs = [os.path.join(ROOT, "pigeon_amd", "libpigeon_hip.so")]
    dev = os.path.join(ROOT, "pigeon_amd", "libpigeon_hip_dev.so")
    if os.path.exists(dev) and os.path.getmtime(dev) >= os.path.getmtime(libs[0]) - 3600:
        libs.append(dev)
    for lib in libs:
        und = subprocess.run(["nm", "-u", "-C", lib], capture_output=True, text=True, check=True).stdout
        bad = [l.strip() for l in und.splitlines() if "_kernel" in l or "__device_stub__" in l]
        assert not bad, f"{lib}: kernels without a host side: {bad[:4]}"
        ctypes.CDLL(lib)


def test_header_is_plain_c():
    """include/pigeon_hip.h is what a cgo / JNI / ctypes binding reads: it must compile as C (no C++, no HIP types), on its own."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        import pytest
        pytest.skip("no C compiler")
    r = subprocess.run([cc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "pigeon_hip.h")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_graft_entry_build_runs():
    """The driver's build check (`__graft_entry__.build()`, run on a box without a GPU every round): compiles / reuses the in-tree
    library, imports the package, checks the ABI version against the header's -- it must not lag behind a version bump."""
    import re
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    header = open(os.path.join(ROOT, "include", "pigeon_hip.h")).read()
    want = int(re.search(r"#define PG_ABI_VERSION\s+(\d+)", header).group(1))
    from pigeon_amd import _lib
    assert _lib.load().pg_abi_version() == want
