"""development: for every VMEM instruction INSIDE an inline-asm block of one kernel of a hipcc -S listing, the wait states
between it and the last VALU write (v_readlane / v_readfirstlane / v_cmp ...) of a scalar register it reads (address base,
M0 source is SALU: not a hazard).  The ISA wants 5 wait states; hipcc pads only its own instructions.
  usage: isa_asm_sgpr_audit.py file.s 'kernel-name-substring (demangled)' [--list]"""
import re
import subprocess
import sys
from collections import Counter

sys.path.insert(0, __import__("os").path.dirname(__file__))


def kernel_body(path, needle):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and ":" in l:
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


SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]|\b(vcc)\b")


def sregs(tok):
    out = set()
    for m in SREG.finditer(tok):
        if m.group(1):
            out.add(int(m.group(1)))
        elif m.group(2):
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.update((106, 107))
    return out


def main():
    body = kernel_body(sys.argv[1], sys.argv[2])
    insts, in_asm = [], False
    for l in body:
        t = l.strip()
        if "#ASMSTART" in t:
            in_asm = True
            continue
        if "#ASMEND" in t:
            in_asm = False
            continue
        if not t or t.startswith(";"):
            continue
        if t.startswith("."):
            if t.endswith(":"):
                insts.append((t, True, False))
            continue
        t = t.split(";")[0].strip()
        if t:
            insts.append((t, False, in_asm))
    hist, shown = Counter(), 0
    for i, (t, lab, asm) in enumerate(insts):
        if lab or not asm:
            continue
        op = t.split()[0]
        if not op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            continue
        ops = t.split(None, 1)[1]
        need = sregs(ops)
        if not need:
            continue
        ws, j, found = 0, i - 1, None
        while j >= 0 and not insts[j][1] and ws < 6:
            tt = insts[j][0]
            o = tt.split()[0]
            if o.startswith("v_") and " " in tt:
                dst = tt.split(None, 1)[1].split(",")[0]
                if sregs(dst) & need:
                    found = (ws, o)
                    break
            if o.startswith("s_") and not o.startswith(("s_nop", "s_waitcnt", "s_barrier", "s_cbranch", "s_branch")) and " " in tt:
                dst = tt.split(None, 1)[1].split(",")[0]
                need = need - sregs(dst)          # an SALU write in between: that is the producer (no hazard)
            m = re.match(r"s_nop\s+(\d+)", tt)
            ws += int(m.group(1)) + 1 if m else 1
            j -= 1
        if found:
            hist[(found[0], found[1].replace("_e32", "").replace("_e64", ""), op)] += 1
            if "--list" in sys.argv and found[0] < 5 and shown < 6:
                shown += 1
                print("--- wait states", found[0])
                for k in range(max(0, i - 8), i + 2):
                    print("      ", insts[k][0])
    n_vmem = sum(1 for t, lab, a in insts if a and not lab and t.split()[0].startswith(("global_", "buffer_")))
    print("VMEM instructions inside asm blocks:", n_vmem)
    print("(wait states since a VALU write of a scalar operand, producer, consumer) -> count   [< 5 is a hazard]")
    for k in sorted(hist):
        print("   ", k, hist[k], "   <-- HAZARD" if k[0] < 5 else "")
    if not hist:
        print("    none within 6 wait states")


main()
