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


# ---- MFMA result hazards -------------------------------------------------------------------------------------------------
# The persistent GEMMs issue their MFMAs as inline asm with the accumulator tied (common.h mfma16_acc / mfma16_init) because
# hipcc's own register allocation of the 4-pass form spills.  hipcc pads NO hazards for instructions inside an asm string, so
# the distance between an MFMA and the first instruction that touches its destination registers (other than the next MFMA
# of the accumulate chain taking it whole as C) is the author's job: `s_nop 7; s_nop 7` after the last phase, barriers and
# unrelated instructions elsewhere.  This pass measures that distance in the final ISA: every v_mfma, the following
# instructions until `passes + 4` wait states have gone by (an N-pass XDL op needs N + 4 states before its D may be read or
# overwritten by anything but a chained MFMA on gfx950: 8-pass -> 12, cdna_hip_programming.md 5.7 item 2; 4-pass -> 8), and
# any register overlap inside that window is reported.  Straight-line scan: past a conditional branch the fall-through path is
# followed (the loop exit into the epilogue), a taken branch is not.
_PASSES = (("32x32x16", 8), ("16x16x32", 4), ("32x32x8", 16), ("16x16x16", 8), ("32x32x4", 16), ("16x16x4", 8), ("4x4x4", 2))
_REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def _regs(operand_text):
    out = set()
    for m in _REG.finditer(operand_text):
        lo = int(m.group(2) if m.group(2) is not None else m.group(4))
        hi = int(m.group(3) if m.group(3) is not None else m.group(4))
        out.update((m.group(1), r) for r in range(lo, hi + 1))
    return out


def _states(op, operands):
    if op == "s_nop":
        try:
            return int(operands.strip(), 0) + 1
        except ValueError:
            return 1
    return 1


def mfma_hazards(dis, kernels=None):
    """{kernel: [(mfma line, offending line, states in between)]} from llvm-objdump -d text."""
    out = {}
    cur, body = None, []

    def flush():
        if cur is None:
            return
        bad = []
        for i, (op, ops, raw) in enumerate(body):
            if not op.startswith("v_mfma") and not op.startswith("v_smfmac"):
                continue
            passes = next((p for k, p in _PASSES if k in op), 8)
            need = passes + 4
            parts = ops.split(",")
            dst = _regs(parts[0])
            gone = 0
            for op2, ops2, raw2 in body[i + 1:]:
                if gone >= need:
                    break
                if op2 in ("s_branch", "s_endpgm", "s_setpc_b64"):
                    break                                        # (a conditional branch: the fall-through path is scanned on)
                touched = _regs(ops2)
                if touched & dst:
                    p2 = ops2.split(",")
                    chained = (op2.startswith("v_mfma") and len(p2) >= 4 and _regs(p2[0]) == dst and _regs(p2[3]) == dst
                               and not (_regs(p2[1]) | _regs(p2[2])) & dst)
                    if not chained:
                        bad.append((raw, raw2, gone))
                        break
                gone += _states(op2, ops2)
        if bad:
            out[cur] = bad

    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            flush()
            cur = m.group(1) if (kernels is None or m.group(1) in kernels) else None
            body = []
            continue
        if cur is None:
            continue
        t = line.strip()
        if not t:
            continue
        t = t.split("//")[0].strip()
        sp = t.split(None, 1)
        body.append((sp[0], sp[1] if len(sp) > 1 else "", t))
    flush()
    return out


def audit_hazards(lib_path=None):
    """MFMA result hazards of every kernel in the library: {mangled kernel: [(mfma, offender, states)]}."""
    lib_path = lib_path or os.path.join(ROOT, "pigeon_amd", "libpigeon_hip.so")
    res = {}
    for co in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
            path = f.name
        try:
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(path)
        res.update(mfma_hazards(dis))
    return res


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
    hz = audit_hazards(args[0] if args else None)
    names = _demangle(list(hz))
    print(f"MFMA result hazards (destination touched inside passes + 4 wait states): {sum(len(v) for v in hz.values())} in {len(hz)} kernel(s)")
    for k, v in hz.items():
        print("  ", names[k][:140])
        for mf, off, gone in v[:4]:
            print(f"      {mf}\n        -> after {gone} state(s): {off}")


if __name__ == "__main__":
    main()
