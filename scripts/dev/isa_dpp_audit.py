"""development: audit of the instructions that feed v_mov_b32_dpp in one kernel of a hipcc -S listing.
For every DPP move: the instruction that last wrote its source register (walking back inside the basic block), the number
of wait states in between (an instruction = 1, s_nop N = N + 1), and what sits between an EXEC write and the move.
  usage: isa_dpp_audit.py file.s 'kernel-name-substring (demangled)' [--list]"""
import re
import subprocess
import sys
from collections import Counter


def kernel_body(path, needle):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and l.rstrip().endswith(":") or (l.startswith("_Z") and ": " in l):
            name = l.split(":")[0]
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout
            if needle in dem:
                start = i
                break
    assert start is not None, "kernel not found"
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l)
    return body


REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for r in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), r))
    return out


def main():
    path, needle = sys.argv[1], sys.argv[2]
    body = kernel_body(path, needle)
    insts = []  # (text, is_label, in_asm)
    in_asm = False
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") and "ASM" not in t:
            continue
        if "#ASMSTART" in t:
            in_asm = True
            continue
        if "#ASMEND" in t:
            in_asm = False
            continue
        if t.startswith("."):
            if t.endswith(":"):
                insts.append((t, True, False))
            continue
        t = t.split(";")[0].strip()
        if t:
            insts.append((t, False, in_asm))
    hist = Counter()
    exec_hist = Counter()
    worst = []
    for i, (t, lab, asm) in enumerate(insts):
        if lab or "_dpp" not in t:
            continue
        ops = t.split(None, 1)[1].split(",")
        src = regs(ops[1])
        ws, j, found, exec_ws = 0, i - 1, None, None
        while j >= 0 and not insts[j][1] and ws < 12:
            tt = insts[j][0]
            op = tt.split()[0]
            if exec_ws is None and re.search(r"\bexec\b", tt.split(None, 1)[1].split(",")[0] if " " in tt else "") and op.startswith(("s_", "v_cmpx")):
                exec_ws = (ws, op, insts[j][2])
            if op.startswith(("v_", "ds_", "global_", "buffer_", "scratch_")) and not op.startswith(("global_store", "ds_write", "scratch_store", "buffer_store")):
                dst = tt.split(None, 1)[1].split(",")[0] if " " in tt else ""
                if regs(dst) & src and found is None:
                    found = (ws, op, insts[j][2])
            m = re.match(r"s_nop\s+(\d+)", tt)
            ws += int(m.group(1)) + 1 if m else 1
            j -= 1
        if found:
            hist[(found[0], found[1].split("_e64")[0], "asm" if found[2] else "")] += 1
            if found[0] < 3:
                worst.append((i, found, t))
        if exec_ws:
            exec_hist[exec_ws] += 1
    print("DPP moves:", sum(1 for t, lab, a in insts if not lab and "_dpp" in t))
    print("producer of the DPP source within 12 wait states: (wait states between, opcode, inside asm) -> count")
    for k in sorted(hist):
        print("   ", k, hist[k])
    print("EXEC write within 12 wait states ahead of a DPP move:")
    for k in sorted(exec_hist):
        print("   ", k, exec_hist[k])
    if "--list" in sys.argv:
        for i, f, t in worst[:40]:
            print("---", f)
            for k in range(max(0, i - 6), i + 1):
                print("      ", insts[k][0])


main()
