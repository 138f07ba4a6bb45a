"""Build libpigeon_hip.so (gfx950) in-tree with hipcc.  `python -m pigeon_amd.build [--force]`.

hipcc cross-compiles without a GPU, so this runs in the authoring container; the resulting .so travels to the
GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libpigeon_hip.so")
# the product library: production kernels only
SOURCES = ["vit.hip", "gemm_bf16.hip", "gemm_pp.hip", "gemm_pp6.hip", "gemm_tail.hip", "gemm_mid.hip", "attention.hip", "rowops.hip", "precise.hip",
           "preprocess.hip", "geo_proto.hip", "head.hip", "refine.hip", "certainty.hip", "requeue.hip", "comm.hip"]
# additionally in the tools build (--dev), from tools/csrc/: kernel generations the product superseded, kept for A/B work
# (gemm variant 64 = gemm_w4.hip; attention variants 1, 4..15 = attention_old.hip)
DEV_DIR = os.path.join(os.path.dirname(HERE), "tools", "csrc")
# (round 5: the two files moved to the branch archive/kernel-generations-r04; a checkout that has them under tools/csrc/ again gets
# them compiled in with -DPIGEON_OLD_GENERATIONS)
DEV_SOURCES = [f for f in (os.path.join(DEV_DIR, "gemm_w4.hip"), os.path.join(DEV_DIR, "attention_old.hip")) if os.path.exists(f)]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "pigeon_internal.h"), os.path.join(CSRC, "gemm_epi.h"),
           os.path.join(CSRC, "attention_common.h"),
           os.path.join(os.path.dirname(HERE), "include", "pigeon_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, dev: bool = False) -> str:
    """dev=False: the product library libpigeon_hip.so -- production kernels only.
    dev=True:  libpigeon_hip_dev.so, compiled with -DPIGEON_ABLATIONS: additionally the superseded kernel generations, A/B arms
               and the timing-only ablation kernels (which compute WRONG results by construction) that tools/ selects through
               PIGEON_GEMM_VARIANT / PIGEON_ATTN_VARIANT; load it with PIGEON_HIP_LIB=pigeon_amd/libpigeon_hip_dev.so."""
    global OBJ, LIB
    obj_dir = os.path.join(CSRC, "build_dev" if dev else "build")
    lib = os.path.join(HERE, "libpigeon_hip_dev.so" if dev else "libpigeon_hip.so")
    flags = FLAGS + (["-DPIGEON_ABLATIONS"] if dev else []) + (["-DPIGEON_OLD_GENERATIONS"] if dev and len(DEV_SOURCES) == 2 else [])
    return _build(obj_dir, lib, flags, force, verbose, SOURCES + (DEV_SOURCES if dev else []))


def _build(OBJ: str, LIB: str, FLAGS, force: bool, verbose: bool, SOURCES) -> str:
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    jobs = []
    for src in SOURCES:
        s = src if os.path.isabs(src) else os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.basename(src).replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [cc] + FLAGS + ["-I", CSRC] + (["-ffp-contract=off"] if s.endswith("preprocess.hip") else []) + ["-c", s, "-o", o]
        if verbose:
            print("[pigeon_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 2)) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s).replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print("[pigeon_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def build_variant(name: str, defines, verbose: bool = True) -> str:
    """A/B arm of the PRODUCT library with extra -D defines (kernel knobs that are compile-time constants, e.g. -DPG_P6_EARLY=8):
    libpigeon_hip_<name>.so; select it with PIGEON_HIP_LIB.  `python -m pigeon_amd.build --variant e8 -DPG_P6_EARLY=8`."""
    return _build(os.path.join(CSRC, f"build_{name}"), os.path.join(HERE, f"libpigeon_hip_{name}.so"), FLAGS + list(defines), False,
                  verbose, SOURCES)


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")]))
    else:
        print(build(force="--force" in sys.argv, dev="--dev" in sys.argv))
