#!/usr/bin/env python3
"""Scan the gfx950 code objects of a built library for the code shape behind round 5/6's fault of the counter build's
k_lsd_grow_mw16 (profiles/r06_prof_build_mw16_fault_root_cause.txt): a vector register written BETWEEN two consecutive EXEC restores
of one basic block,

        s_and_saveexec_b64 s[0:1], vcc     ; if (...)
        s_cbranch_execz JOIN
        ...                                ;   (a divergent loop; its own restore s_or_b64 exec, exec, s[2:3] ends the body)
  JOIN: v_mov_b32 v46, 1                   ; <- a value the register allocator re-materialised here (live-range split around the call
        s_or_b64 exec, exec, s[0:1]        ;    below): only the lanes of the `if` get it -- none when the branch was taken
        ...
        s_swappc_b64 ...
        v_mov_b32 v35, v46                 ; copied back in ALL lanes: the lanes that were off hold whatever v46 held before

-- ROCm 7.2's register allocator placed the split copy of a loop-invariant constant (the `1` region growing stores as a pixel's
private mark) above the second restore; lanes that were inactive there stored a stale byte as their mark, an even byte reads back
as "not marked", and the region grew for ever (until its queue ran off the end of the workspace: the memory fault).

What is reported: every vector-register write that stands at the top of a JOIN block -- the target of an s_cbranch_execz, entered
with EXEC = 0 when the branch is taken and with the inner region's lanes otherwise -- in front of that block's EXEC restore
(s_or_b64 exec, exec, s[..]) and whose destination is read behind the restore.  Code the source put there does not exist (the
compiler has no reason to compute something for all lanes in front of the restore that brings them back); an else-branch
(s_andn2_saveexec / s_or_saveexec) or a nested region ends the scan.  tests/test_kernel_resources.py asserts that the shipped
library AND the counter build have none.

    python tools/isa_exec_split_check.py [pl-slam_amd/libplslam_hip.so]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import LLVM, MAGIC, TARGET  # noqa: E402

INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
RESTORE = re.compile(r"^s_or_b64 exec, exec, s\[\d+:\d+\]$|^s_mov_b64 exec, s\[\d+:\d+\]$|^s_or_saveexec_b64")
MOVES = ("v_mov_b32_e32", "v_mov_b64_e32", "v_mov_b32_e64", "v_accvgpr_write_b32", "v_accvgpr_read_b32")


def code_objects(lib, td):
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(td, "copy.so")])
    d = open(fat, "rb").read()
    pos = [m.start() for m in re.finditer(re.escape(MAGIC), d)]
    for i, p in enumerate(pos):
        e = pos[i + 1] if i + 1 < len(pos) else len(d)
        b, co = os.path.join(td, "b%d.bin" % i), os.path.join(td, "co%d.o" % i)
        open(b, "wb").write(d[p:e])
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + b, "--targets=" + TARGET,
                            "--output=" + co], capture_output=True)
        if r.returncode == 0 and os.path.exists(co):
            yield co


def regs(tok):
    """VGPR numbers named by an operand token: v7, v[4:5], |v3|, -v[8:9]."""
    out = []
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def functions(co):
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout
    name, ins = None, []
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m:
            if not re.match(r"^(lsdwalk|L)\w*$", m.group(1)) and "+" not in m.group(1):   # (inline-asm labels are not functions)
                if name and ins:
                    yield name, ins
                name, ins = m.group(1), []
            continue
        m = INS.match(ln)
        if m and name:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if name and ins:
        yield name, ins


def branch_target(a, args):
    try:
        off = int(args.split()[0])
    except (ValueError, IndexError):
        return None
    off = off - 65536 if off >= 32768 else off
    return a + 4 + 4 * off


def scan(ins):
    """Vector writes at the top of a join block -- the target of an s_cbranch_execz, reached with EXEC = 0 when the branch is taken
    and with the inner region's lanes otherwise -- that stand IN FRONT of the block's EXEC restore and whose destination is read
    behind it."""
    index = {a: k for k, (a, _, _) in enumerate(ins)}
    leaders, joins = set(), set()
    for a, op, args in ins:
        if op.startswith("s_cbranch") or op == "s_branch":
            t = branch_target(a, args)
            if t is not None:
                leaders.add(t)
                if op == "s_cbranch_execz":
                    joins.add(t)
    found = []
    for t in sorted(joins):
        if t not in index:
            continue
        j, mid = index[t], []
        n = len(ins)
        while j < n:
            aj, oj, gj = ins[j]
            if (j > index[t] and aj in leaders) or oj.startswith("s_cbranch") or oj in ("s_branch", "s_swappc_b64", "s_setpc_b64", "s_endpgm"):
                j = n
                break
            if RESTORE.match((oj + " " + gj).strip()):
                break
            if "saveexec" in oj or gj.split(",")[0].strip() == "exec":   # an else / a nested region begins: its code is not the join's
                j = n
                break
            mid.append(ins[j])
            j += 1
        if j >= n or not mid:
            continue
        for am, om, gm in mid:
            if not om.startswith("v_") or om.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_writelane", "v_nop")):
                continue
            dst = regs(gm.split(",")[0])
            if not dst:
                continue
            k, live, used = j + 1, set(dst), False
            while k < n and live:
                ak, ok, gk = ins[k]
                if ak in leaders or ok.startswith("s_cbranch") or ok in ("s_branch", "s_setpc_b64", "s_endpgm"):
                    break
                parts = gk.split(",")
                stores = ok.startswith(("global_store", "scratch_store", "flat_store", "ds_write", "buffer_store"))
                reads = regs(gk if stores or ok.startswith(("v_cmp", "s_", "v_readlane", "v_readfirstlane")) else ",".join(parts[1:]))
                if live & set(reads):
                    used = True
                    break
                if not stores:
                    live -= set(regs(parts[0]))
                k += 1
            if used:
                found.append((am, om + " " + gm, t, ins[j][0]))
    return found


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pl-slam_amd", "libplslam_hip.so")
    total = 0
    with tempfile.TemporaryDirectory() as td:
        for co in code_objects(lib, td):
            for name, ins in functions(co):
                hits = scan(ins)
                short = re.sub(r"^_ZN3plh\d+", "", name)[:40]
                for a, text, r0, r1 in hits:
                    print("%-40s +0x%05x  %-44s (join block at +0x%x, its EXEC restore at +0x%x)" % (short, a - ins[0][0], text, r0 - ins[0][0], r1 - ins[0][0]))
                total += len(hits)
    print("%d vector writes in front of a join block's EXEC restore whose value is read behind it" % total)
    return 0


if __name__ == "__main__":
    sys.exit(main())
