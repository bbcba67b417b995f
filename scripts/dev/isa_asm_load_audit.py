"""development: inline-asm LDS / global loads whose destination registers are touched before the data has landed.
hipcc counts an asm load's destination as written at ;;#ASMEND: under register pressure it may copy (v_mov, v_accvgpr_write),
spill or reuse the register BEFORE the hand-written s_waitcnt that covers the load.  The audit walks one kernel of a
hipcc -S listing, keeps the queue of outstanding LDS operations (every ds_* instruction counts for lgkmcnt, returns in
order) and reports every instruction that reads or writes a destination of an asm ds_read that is still in the queue.
  usage: isa_asm_load_audit.py file.s 'kernel-name-substring (demangled)' [--list]"""
import re
import subprocess
import sys


def kernel_body(path, needle):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and ":" in l:
            dem = subprocess.run(["c++filt", l.split(":")[0]], capture_output=True, text=True).stdout
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
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def main():
    body = kernel_body(sys.argv[1], sys.argv[2])
    in_asm, queue, bad, n_loads = False, [], [], 0   # queue: (dest regs or empty set, text)
    for ln, l in enumerate(body):
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
                queue = []                       # basic-block boundary: start afresh (the loops re-enter with waits)
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        op = t.split()[0]
        ops = t.split(None, 1)[1] if " " in t else ""
        m = re.match(r"s_waitcnt.*lgkmcnt\((\d+)\)", t)
        if m:
            n = int(m.group(1))
            queue = queue[len(queue) - n:] if n and len(queue) > n else ([] if not n else queue)
            continue
        if op == "s_waitcnt" and "lgkmcnt" not in t and "vmcnt" not in t and "expcnt" not in t:
            queue = []                           # (numeric form: treat as a full wait)
            continue
        pending = set().union(*[q[0] for q in queue]) if queue else set()
        if pending and not in_asm:
            touched = regs(ops) & pending
            if touched:
                bad.append((ln, t, sorted(touched), [q[1] for q in queue if q[0] & touched][0]))
        if op.startswith("ds_"):
            if op.startswith("ds_read") and in_asm:
                n_loads += 1
                queue.append((regs(ops.split(",")[0]), t))
            else:
                queue.append((set(), t))
    print("asm LDS loads:", n_loads, "  instructions touching a destination before its wait:", len(bad))
    for ln, t, regs_, src in bad[: (40 if "--list" in sys.argv else 8)]:
        print("   line %d: %-60s  touches %s of  %s" % (ln, t, regs_, src))


main()
