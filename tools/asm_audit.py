"""Static audit of the gfx950 code objects inside libpigeon_hip.so: registers, scratch (spills) and WATERFALL LOOPS per kernel.

   python tools/asm_audit.py [path/to/libpigeon_hip.so] [--all]

A waterfall loop is what hipcc emits when a buffer descriptor (or an SGPR-only operand such as a buffer instruction's soffset)
lives in VGPRs: `v_readfirstlane_b32` x N, `v_cmp_eq`, `s_and_saveexec_b64`, the memory instruction, `s_cbranch_execnz` back.
In round 2 one clamp (`min / max` -> `v_med3_i32`, there is no scalar med3) put the record count of the row-statistics descriptor
into a VGPR and wrapped every statistics load of the QKV / fc1 epilogues in such a loop: 2 % of the benchmark, invisible in any
profile short of reading the ISA.  tests/test_asm_audit.py keeps the persistent kernels free of them (and of spills beyond 16 bytes).

No GPU needed: the bundles are cut out of the library's .hip_fatbin by hand (clang offload bundle format: magic, entry count,
{offset, size, id length, id}), disassembled with /opt/rocm/lib/llvm/bin/llvm-objdump, metadata from llvm-readelf --notes.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path, arch="gfx950"):
    """Every device code object (bytes) for `arch` bundled into the shared library."""
    blob = open(lib_path, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), blob):
        base = m.start()
        p = base + len(MAGIC)
        (n,) = struct.unpack_from("<Q", blob, p)
        p += 8
        if n > 16:
            continue                                             # the magic string inside some other data
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, p)
            p += 24
            ident = blob[p:p + idlen].decode("ascii", "replace")
            p += idlen
            if arch in ident and size > 0:
                out.append(blob[base + off: base + off + size])
    return out


def _demangle(names):
    try:
        r = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True, check=True)
        return dict(zip(names, r.stdout.split("\n")))
    except Exception:
        return {n: n for n in names}


def audit_code_object(co_bytes):
    """{mangled kernel name: {vgpr, agpr, sgpr, scratch, waterfall, instructions}} of one code object."""
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(co_bytes)
        path = f.name
    try:
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True, check=True).stdout
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
    finally:
        os.unlink(path)
    kernels = {}
    # metadata: one YAML-ish block per kernel; the fields we want are flat `.key: value` lines
    for blk in re.split(r"\n\s*- \.agpr_count:", "\n" + notes)[1:]:
        blk = ".agpr_count:" + blk
        def field(k, d=0):
            m = re.search(r"\." + k + r":\s*(\S+)", blk)
            return m.group(1) if m else d
        name = field("name", None)
        if not name:
            continue
        kernels[name] = {"vgpr": int(field("vgpr_count")), "agpr": int(field("agpr_count")), "sgpr": int(field("sgpr_count")),
                         "scratch": int(field("private_segment_fixed_size")), "waterfall": 0, "instructions": 0}
    cur, window = None, []
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1) if m.group(1) in kernels else None
            window = []
            continue
        if cur is None:
            continue
        ins = line.strip().split()
        if not ins:
            continue
        op = ins[0]
        kernels[cur]["instructions"] += 1
        window.append(op)
        if len(window) > 24:
            window.pop(0)
        if op == "s_cbranch_execnz" and "v_readfirstlane_b32" in window and any(w.startswith("s_and_saveexec") for w in window):
            kernels[cur]["waterfall"] += 1
            window = []
    return kernels


def audit(lib_path=None):
    lib_path = lib_path or os.path.join(ROOT, "pigeon_amd", "libpigeon_hip.so")
    res = {}
    for co in code_objects(lib_path):
        res.update(audit_code_object(co))
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    res = audit(args[0] if args else None)
    pretty = _demangle(list(res))
    show_all = "--all" in sys.argv
    print(f"{'vgpr':>5} {'agpr':>5} {'scratch':>8} {'waterfall':>10} {'instr':>7}  kernel")
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["instructions"]):
        if show_all or v["scratch"] or v["waterfall"] or v["instructions"] > 1500:
            print(f"{v['vgpr']:5d} {v['agpr']:5d} {v['scratch']:8d} {v['waterfall']:10d} {v['instructions']:7d}  {pretty[k][:150]}")


if __name__ == "__main__":
    main()
